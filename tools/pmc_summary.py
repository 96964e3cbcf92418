"""Per-kernel HBM traffic from two separate rocprofv3 counter passes (the MI355X_MICROARCH.md recipe):

  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d <dirF> -o f -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d <dirW> -o w -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events
  python tools/pmc_summary.py <dirF> <dirW> > profiles/r01_pmc_traffic.txt

FETCH_SIZE / WRITE_SIZE are in KB; gfx950 reports half of wide coalesced reads, hence FETCH x2."""
import csv, glob, os, re, sys
from collections import defaultdict


def collect(d, counter):
    acc = defaultdict(lambda: [0, 0.0])
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            rd = csv.DictReader(f)
            per_dispatch = defaultdict(float)
            names = {}
            for row in rd:
                if row.get("Counter_Name") != counter:
                    continue
                key = (row.get("Dispatch_Id"), row.get("Agent_Id"))
                per_dispatch[key] += float(row["Counter_Value"])
                names[key] = row["Kernel_Name"]
            for key, v in per_dispatch.items():
                a = acc[names[key]]
                a[0] += 1
                a[1] += v
    return acc


def short(n):
    n = n.replace("(anonymous namespace)::", "")
    n = re.sub(r"^void ", "", n)
    return n[:70]


def main():
    dF, dW = sys.argv[1], sys.argv[2]
    F, W = collect(dF, "FETCH_SIZE"), collect(dW, "WRITE_SIZE")
    print("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --output-format csv) -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events")
    print("per-launch averages; FETCH_SIZE/WRITE_SIZE are in KB; gfx950 reports 1/2 of wide coalesced reads -> FETCH x2 (MI355X_MICROARCH.md, HBM/rocprofv3 section)")
    print("kernel | launches sampled | FETCH_SIZE KB | WRITE_SIZE KB | corrected HBM MB per launch")
    rows = []
    for k, (n, v) in F.items():
        wn, wv = W.get(k, (0, 0.0))
        f_avg = v / max(n, 1)
        w_avg = wv / max(wn, 1)
        rows.append((n * (2 * f_avg + w_avg), k, n, f_avg, w_avg))
    for _, k, n, f_avg, w_avg in sorted(rows, reverse=True)[:24]:
        print(f"{short(k)} | {n} | {f_avg:.1f} | {w_avg:.1f} | {(2 * f_avg + w_avg) / 1024.0:.2f}")


if __name__ == "__main__":
    main()
