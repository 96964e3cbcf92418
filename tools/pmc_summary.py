"""Per-kernel HBM traffic from two separate rocprofv3 counter passes (the MI355X_MICROARCH.md recipe):

  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d <dirF> -o f -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d <dirW> -o w -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events
  python tools/pmc_summary.py <dirF> <dirW> > profiles/r01_pmc_traffic.txt

FETCH_SIZE / WRITE_SIZE are in KB; gfx950 reports half of wide coalesced reads, hence FETCH x2.
With a third argument "grid" the rows are per (kernel, grid size): one row per layer shape when the traced program is
tools/bench_conv.py."""
import csv, glob, os, re, sys
from collections import defaultdict


BY_GRID = False


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
                key = (int(row.get("Dispatch_Id")), row.get("Agent_Id"))
                per_dispatch[key] += float(row["Counter_Value"])
                names[key] = row["Kernel_Name"] + (f" grid {row.get('Grid_Size', '?')}" if BY_GRID else "")
            if BY_GRID:          # consecutive launches of one (kernel, grid) = one timed set of the traced program: number the sets
                run, prev = 0, None
                for key in sorted(per_dispatch):
                    if names[key] != prev:
                        run, prev = run + 1, names[key]
                    names[key] = f"{names[key]} set {run:03d}"
            for key, v in per_dispatch.items():
                a = acc[names[key]]
                a[0] += 1
                a[1] += v
    return acc


def short(n):
    n = n.replace("(anonymous namespace)::", "")
    n = re.sub(r"^void ", "", n)
    if BY_GRID:
        return n[n.rindex(" set") + 1:] + " " + n.split("(")[0][:48] + " " + n[n.rindex(" grid"):n.rindex(" set")]
    return n[:70]


def main():
    global BY_GRID
    dF, dW = sys.argv[1], sys.argv[2]
    BY_GRID = len(sys.argv) > 3 and sys.argv[3] == "grid"
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
    for _, k, n, f_avg, w_avg in (sorted(rows, key=lambda r: r[1][r[1].rindex(' set'):]) if BY_GRID else sorted(rows, reverse=True)[:24]):
        print(f"{short(k)} | {n} | {f_avg:.1f} | {w_avg:.1f} | {(2 * f_avg + w_avg) / 1024.0:.2f}")


if __name__ == "__main__":
    main()
