"""Summarise rocprofv3 --pmc passes for the two gather kernels.

Usage (on the GPU box; one pass per counter, kernel-trace only):
    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python bench.py ...
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python bench.py ...
    python tools/pmc_gather.py $OUT/fetch $OUT/write > profiles/rNN_pmc_gather.json

Counter_Value of FETCH_SIZE / WRITE_SIZE is in KiB.  WRITE_SIZE matches the
algorithmic writes exactly; FETCH_SIZE under-reports streaming reads on gfx950
(MI355X_MICROARCH.md, HBM section) and is calibrated on the smallest k_batch_states_u8
launch (the acting gather: every frame of the launch is distinct, so its bytes are known).
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

FRAME = 84 * 84
TILES = -(-FRAME // 1024)


def load(d, counter):
    out = defaultdict(list)
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                if r.get("Counter_Name") != counter:
                    continue
                name = r["Kernel_Name"]
                if "k_batch_experiences" in name:
                    kind = "k_batch_experiences"
                elif "k_batch_states_u8_raw" in name:
                    kind = "k_batch_states_u8_raw"      # (u8 NHWC4 out: one byte per frame byte)
                elif "k_batch_states_u8" in name:
                    kind = "k_batch_states_u8"
                else:
                    continue
                blocks = int(r["Grid_Size"]) // int(r["Workgroup_Size"])
                # normalise the grid to "frame workgroups": the channels-last kernels
                # launch TILES (ceil(7056 / 1024) = 7) workgroups per 4-frame observation
                if kind == "k_batch_states_u8_raw":
                    blocks = (blocks // TILES) * 4     # (7 tiles of 256 dwords per observation)
                elif "nhwc4" in name:
                    if kind == "k_batch_experiences":
                        blocks = (blocks // (2 * TILES)) * 8 + 1
                    else:
                        blocks = (blocks // TILES) * 4
                out[(kind, blocks)].append(float(r["Counter_Value"]))
    return out


def main_sac(fetch_dir, write_dir, entries, cal):
    """configs[4] (vector observations: the fused gather is plain f32 row copies, no u8 frames to
    calibrate FETCH_SIZE on): the dominant k_batch_experiences launch shape of the run, ``entries``
    sampled transitions per launch (bench.py's roofline.entries_per_launch), FETCH_SIZE corrected
    by the factor ``cal`` calibrated on the same box by the DQN passes (rNN_pmc_gather.json)."""
    import hashlib

    def dominant(d, counter):
        by = defaultdict(list)
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(path) as f:
                for r in csv.DictReader(f):
                    if r.get("Counter_Name") == counter and "k_batch_experiences" in r["Kernel_Name"]:
                        by[int(r["Grid_Size"])].append(float(r["Counter_Value"]))
        grid = max(by, key=lambda g: len(by[g]) * g)
        return grid, by[grid]

    gf, f = dominant(fetch_dir, "FETCH_SIZE")
    gw, w = dominant(write_dir, "WRITE_SIZE")
    assert gf == gw, (gf, gw)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    for rel in ("pfrl_amd/csrc/replay.hip", "pfrl_amd/csrc/nhwc.h", "pfrl_amd/csrc/common.h"):
        with open(os.path.join(root, rel), "rb") as fh:
            h.update(fh.read())
    per = 2 * (2 * 376 * 4) + 2 * 17 * 4 + 2 * 12          # bench.py compute_roofline, sac
    fk, wk = sum(f) / len(f), sum(w) / len(w)
    rb, wb = fk * 1024 * cal, wk * 1024
    item = {"launches": len(f), "grid_threads": gf, "algorithmic_read_B": entries * per // 2,
            "algorithmic_write_B": entries * per // 2, "FETCH_SIZE_KiB": round(fk, 2),
            "WRITE_SIZE_KiB": round(wk, 2), "fetch_bytes_corrected": int(rb), "write_bytes": int(wb),
            "traffic_bytes_per_launch": int(rb + wb), "algorithmic_bytes_per_launch": entries * per,
            "traffic_over_algorithmic": round((rb + wb) / (entries * per), 4)}
    json.dump({"unit_note": "FETCH_SIZE / WRITE_SIZE in KiB; FETCH_SIZE corrected by the factor calibrated on "
                            "the u8 acting gather of the DQN passes of the same collection",
               "kernel_sources_sha16": h.hexdigest()[:16], "fetch_calibration_factor": cal,
               "kernels": {"k_batch_experiences (%d entries)" % entries: item}}, sys.stdout, indent=1)


def main():
    if len(sys.argv) > 3 and sys.argv[3] == "--sac":
        return main_sac(sys.argv[1], sys.argv[2], int(sys.argv[4]), float(sys.argv[5]))
    fetch = load(sys.argv[1], "FETCH_SIZE")
    write = load(sys.argv[2], "WRITE_SIZE")
    import hashlib

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    for rel in ("pfrl_amd/csrc/replay.hip", "pfrl_amd/csrc/nhwc.h", "pfrl_amd/csrc/common.h"):
        with open(os.path.join(root, rel), "rb") as f:
            h.update(f.read())
    res = {"unit_note": __doc__.split("Counter_Value")[1].strip().replace("\n", " "),
           # (the build these passes describe: bench.py attaches them only to the same sources)
           "kernel_sources_sha16": h.hexdigest()[:16],
           "kernels": {}}
    # calibration: the smallest acting gather (N observations = 4N frame workgroups, all
    # frames distinct within one launch): known bytes / reported bytes
    cal = None
    for ckind in ("k_batch_states_u8", "k_batch_states_u8_raw"):
        acts = sorted(b for (kind, b) in fetch if kind == ckind)
        if acts:
            key = (ckind, acts[0])
            kib = sum(fetch[key]) / len(fetch[key])
            cal = acts[0] * FRAME / (kib * 1024)
            res["fetch_calibrated_on"] = "%s (%d frames)" % (ckind, acts[0])
            break
    res["fetch_calibration_factor"] = cal
    for (kind, blocks) in sorted(set(fetch) | set(write)):
        f = fetch.get((kind, blocks), [])
        w = write.get((kind, blocks), [])
        if kind == "k_batch_experiences":
            entries = (blocks - 1) // 8          # 2 * B * k frame blocks + scalar blocks
            alg_r, alg_w = entries * 8 * FRAME, entries * 8 * FRAME * 4
            label = "%s (%d entries)" % (kind, entries)
        else:
            frames = blocks
            alg_r, alg_w = frames * FRAME, frames * FRAME * (1 if kind.endswith("_raw") else 4)
            label = "%s (%d frames)" % (kind, frames)
        item = {"launches": max(len(f), len(w)), "algorithmic_read_B": alg_r,
                "algorithmic_write_B": alg_w}
        if f:
            item["FETCH_SIZE_KiB"] = round(sum(f) / len(f), 2)
        if w:
            item["WRITE_SIZE_KiB"] = round(sum(w) / len(w), 2)
        if f and w and cal:
            rb = item["FETCH_SIZE_KiB"] * 1024 * cal
            wb = item["WRITE_SIZE_KiB"] * 1024
            item["fetch_bytes_corrected"] = int(rb)
            item["write_bytes"] = int(wb)
            item["traffic_bytes_per_launch"] = int(rb + wb)
            item["algorithmic_bytes_per_launch"] = alg_r + alg_w
            item["traffic_over_algorithmic"] = round((rb + wb) / (alg_r + alg_w), 4)
        res["kernels"][label] = item
    if len(sys.argv) > 3:     # keep only the kernel named on the command line
        res["kernels"] = {k: v for k, v in res["kernels"].items() if k.startswith(sys.argv[3])}
    json.dump(res, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
