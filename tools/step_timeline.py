"""Where one train step's time is, from a rocprofv3 kernel trace of bench.py:
  cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $REPO/bench.py --steps 6 --warmup 3 \
        --no-cpu-baseline --no-infer --no-kernel-events
  python3 tools/step_timeline.py $OUT [step-from-the-end, default 2]
Prints, for one steady-state step (bounded by two launches of the target-building kernel): the step's length, per queue the busy
time / launches / gaps, the time with no kernel on any queue, the phases (forward = up to the loss kernel, backward = up to the
last weight-gradient kernel, tail), and per phase the kernels of the main queue with their in-step time."""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:44]


def main():
    rows = []
    for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0"),
                         (r.get("Grid_Size_X", "?"), r.get("Grid_Size_Y", "?"))))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if "write_dense_kernel" in r[2]]
    # one step builds targets for three scales: the first launch of each group of marks closer than 1 ms starts a step
    starts = [m for k, m in enumerate(marks) if k == 0 or rows[m][0] - rows[marks[k - 1]][0] > 2_000_000]
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    i0, i1 = starts[-back - 1], starts[-back]
    step = rows[i0:i1]
    t0, t1 = step[0][0], rows[i1][0]
    print(f"step: {(t1 - t0) / 1e6:.3f} ms, {len(step)} launches")
    byq = defaultdict(list)
    for r in step:
        byq[r[3]].append(r)
    main_q = max(byq, key=lambda q: len(byq[q]))
    for q, v in sorted(byq.items(), key=lambda kv: -len(kv[1])):
        busy = sum(e - s for s, e, *_ in v)
        gaps = sum(max(0, v[k + 1][0] - v[k][1]) for k in range(len(v) - 1))
        print(f"queue {q}{' (main)' if q == main_q else ''}: {len(v)} launches, busy {busy / 1e6:.3f} ms, gaps between its launches {gaps / 1e6:.3f} ms, "
              f"first {(v[0][0] - t0) / 1e6:.3f} last end {(v[-1][1] - t0) / 1e6:.3f} ms")
    # time with no kernel anywhere
    ev = sorted((s, e) for s, e, *_ in step)
    cover, cur_s, cur_e = 0, ev[0][0], ev[0][1]
    for s, e in ev[1:]:
        if s > cur_e:
            cover += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    cover += cur_e - cur_s
    print(f"no kernel on any queue: {(t1 - t0 - cover) / 1e6:.3f} ms")
    loss = [r for r in step if "loss_cell" in r[2]]
    wg = [r for r in step if "wgrad" in r[2]]
    t_loss = loss[0][0] if loss else t0
    t_bwd = max(r[1] for r in wg) if wg else t1
    print(f"phases: forward {(t_loss - t0) / 1e6:.3f} ms | backward {(t_bwd - t_loss) / 1e6:.3f} ms | tail {(t1 - t_bwd) / 1e6:.3f} ms")
    for name, lo, hi in (("forward", t0, t_loss), ("backward", t_loss, t_bwd), ("tail", t_bwd, t1)):
        for q, v in sorted(byq.items(), key=lambda kv: -len(kv[1])):
            agg = defaultdict(lambda: [0, 0])
            for s, e, n, *_ in v:
                if lo <= s < hi:
                    agg[short(n)][0] += e - s
                    agg[short(n)][1] += 1
            if not agg:
                continue
            tot = sum(a[0] for a in agg.values())
            print(f"  {name}, queue {q}: busy {tot / 1e6:.3f} ms")
            for n, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:12]:
                print(f"      {t / 1e6:7.3f} ms  {c:3d} x {t / c / 1e3:7.1f} us  {n}")
    for pat in sys.argv[3:]:
        by_grid(step, pat)


def by_grid(step, pat):
    """durations of the launches whose kernel name contains `pat`, grouped by grid size"""
    agg = defaultdict(list)
    for s, e, n, _, g in step:
        if pat in n:
            agg[g].append((e - s) / 1e3)
    print(f"  {pat}: launches by grid")
    for g, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        print(f"      grid {g[0]:>9} x {g[1]:>3}: {len(v):3d} x {sum(v) / len(v):7.1f} us (min {min(v):6.1f})")


if __name__ == "__main__":
    main()
