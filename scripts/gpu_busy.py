"""Share of wall time the GPU is busy in a rocprofv3 kernel trace: sum of kernel durations / (last end - first start) over the
kernels between two marker positions (default: the whole trace), and the largest gaps with the kernels around them.
    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 scripts/att_bench.py ...
    python scripts/gpu_busy.py DIR [--skip-first-s 2.0] [--last-ms 400]"""
import csv
import glob
import sys


def main():
    d = sys.argv[1]
    skip = float(sys.argv[sys.argv.index("--skip-first-s") + 1]) if "--skip-first-s" in sys.argv else 0.0
    f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[-1]
    rows = list(csv.DictReader(open(f)))
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
    t0 = ev[0][0]
    ev = [e for e in ev if (e[0] - t0) * 1e-9 >= skip]
    if "--last-ms" in sys.argv:
        last = float(sys.argv[sys.argv.index("--last-ms") + 1]) * 1e6
        ev = [e for e in ev if ev[-1][1] - e[0] <= last]
    span = ev[-1][1] - ev[0][0]
    busy = sum(e[1] - e[0] for e in ev)
    print(f"{len(ev)} kernels over {span * 1e-6:.1f} ms, busy {busy * 1e-6:.1f} ms = {busy / span:.3f}")
    gaps = sorted(((ev[i + 1][0] - ev[i][1], ev[i][2][:50], ev[i + 1][2][:50]) for i in range(len(ev) - 1)), reverse=True)
    tot_gap = sum(g[0] for g in gaps if g[0] > 0)
    print(f"gaps: total {tot_gap * 1e-6:.1f} ms; > 20 us: {sum(1 for g in gaps if g[0] > 20000)}; the largest:")
    for g in gaps[:12]:
        print(f"  {g[0] * 1e-3:8.1f} us  after {g[1]}  before {g[2]}")


if __name__ == "__main__":
    main()
