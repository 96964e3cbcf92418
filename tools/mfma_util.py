"""Per-kernel MFMA utilisation from one rocprofv3 counter pass (MI355X_MICROARCH.md recipe):

  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d <dir> -o m -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events --no-infer
  python tools/mfma_util.py <dir> > profiles/rNN_mfma_util.txt

utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x elapsed cycles), elapsed cycles = GRBM_GUI_ACTIVE / 8 (the counter is
summed over the 8 XCDs).  Counter collection serialises the kernels: these are SOLO figures."""
import csv, glob, os, re, sys
from collections import defaultdict


def main():
    d = sys.argv[1]
    per = defaultdict(lambda: defaultdict(float))
    meta = {}
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                key = (row["Dispatch_Id"], row.get("Agent_Id"))
                per[key][row["Counter_Name"]] += float(row["Counter_Value"])
                meta[key] = (row["Kernel_Name"], float(row.get("End_Timestamp", 0)) - float(row.get("Start_Timestamp", 0)))
    agg = defaultdict(lambda: [0, 0.0, 0.0, 0.0])
    for key, c in per.items():
        name, ns = meta[key]
        a = agg[name]
        a[0] += 1; a[1] += c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0); a[2] += c.get("GRBM_GUI_ACTIVE", 0.0); a[3] += ns
    print("rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -- python3 bench.py --steps 2 --warmup 1 "
          "--no-cpu-baseline --no-kernel-events --no-infer   (own pass; counter collection serialises the kernels: SOLO figures)")
    print("MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x elapsed cycles), elapsed cycles = GRBM_GUI_ACTIVE / 8; "
          "effective clock = elapsed cycles / kernel wall time")
    print("kernel | launches | avg us | MFMA busy cycles per launch | MFMA utilisation | effective clock GHz")
    rows = []
    for name, (n, busy, gui, ns) in agg.items():
        if busy <= 0:
            continue
        el = gui / 8.0
        short = re.sub(r"^void ", "", name.replace("(anonymous namespace)::", ""))[:70]
        rows.append((busy, f"{short} | {n} | {ns / n / 1e3:.1f} | {busy / n:.3e} | {busy / (1024.0 * el):.3f} | {el / max(ns, 1):.2f}"))
    for _, line in sorted(rows, reverse=True):
        print(line)


if __name__ == "__main__":
    main()
