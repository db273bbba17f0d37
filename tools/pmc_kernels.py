"""Average PMC counters per kernel name from a rocprofv3 --pmc run directory."""
import csv, glob, sys, re
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0].replace("void ", "")[:60]
        if not any(k in n for k in ("k_conv", "k_splitk", "k_linear", "k_rmsprop", "k_dqn")):
            continue
        a = acc[n][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
for n, cs in sorted(acc.items()):
    print(n)
    print("    " + "  ".join("%s=%.0f" % (c, v[0] / max(v[1], 1)) for c, v in sorted(cs.items())) + "  (n=%d)" % max(v[1] for v in cs.values()))
