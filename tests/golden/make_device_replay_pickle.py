#!/usr/bin/env python
"""Writes tests/golden/device_replay_save.{pkl,npz} ON THE GPU BOX: a ``ReplayBuffer.save()`` of
an HBM-resident buffer (device env observations, n-step = 3) and, next to it, the arrays the
file has to contain.  tests/test_replay_buffers.py then loads the .pkl with the REFERENCE's
``ReplayBuffer.load`` in the build container (no pfrl_amd on its path).

    gpurun -- 'python tests/golden/make_device_replay_pickle.py gpurun_out'
    cp gpurun_out/device_replay_save.* tests/golden/
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main(outdir):
    from pfrl_amd import replay_buffers
    from pfrl_amd.device_store import DeviceFrameStore
    from pfrl_amd.envs import SyntheticAtariVectorEnv

    dev = torch.device("cuda:0")
    np.random.seed(0)
    N = 3
    store = DeviceFrameStore(512, (12, 12), torch.uint8, dev, stack=4)
    env = SyntheticAtariVectorEnv(N, store=store, seed=7, n_actions=4, p_done=0.1)
    rbuf = replay_buffers.ReplayBuffer(40, num_steps=3, device=dev)
    obss = env.reset()
    for t in range(30):
        actions = np.random.randint(0, 4, size=N)
        nxt, rs, dones, _ = env.step(actions)
        for i in range(N):
            rbuf.append(state=obss[i], action=int(actions[i]), reward=float(rs[i]),
                        next_state=nxt[i], is_state_terminal=bool(dones[i]), env_id=i)
            if dones[i]:
                rbuf.stop_current_episode(env_id=i)
        obss = env.reset(~np.asarray(dones))
    rbuf.save(os.path.join(outdir, "device_replay_save.pkl"))
    ents = [rbuf.memory[i] for i in range(len(rbuf))]
    np.savez(os.path.join(outdir, "device_replay_save.npz"),
             lens=np.array([len(e) for e in ents]),
             first_state=np.stack([np.asarray(e[0]["state"]) for e in ents]),
             last_next_state=np.stack([np.asarray(e[-1]["next_state"]) for e in ents]),
             actions=np.array([e[0]["action"] for e in ents]),
             rewards=np.array([[t["reward"] for t in e] + [0.0] * (3 - len(e)) for e in ents]),
             terminal=np.array([e[-1]["is_state_terminal"] for e in ents]))
    print("wrote", len(ents), "entries")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else ".")
