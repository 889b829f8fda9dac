#!/usr/bin/env python3
"""Per-queue timeline of a rocprofv3 --kernel-trace run of bench.py: where a batch's time goes beyond its kernels.

  python3 tools/timeline.py TRACE_DIR [N_BATCHES]

Takes the last *kernel_trace.csv under TRACE_DIR, keeps the steady-state tail of the run (the last 40 % of the span between the
first and the last scan kernel), and prints
  * per queue: busy fraction, number of kernels, the kernel families seen;
  * for the queue that runs the scan: N_BATCHES consecutive batches, kernel by kernel -- start offset, duration, gap to the
    previous kernel on the same queue (a gap = launch / dependency latency or a wait for another lane);
  * mean scan-to-scan period, mean sum of kernel time and mean sum of gaps per batch on that queue;
  * for the encoder queues: mean span of a forward (first embed kernel -> pool kernel) and its sum of kernel time.
"""
import collections, csv, glob, os, re, subprocess, sys

csv.field_size_limit(1 << 30)


_dm = {}


def short(name):
    if name not in _dm:
        _dm[name] = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
    name = _dm[name]
    m = re.search(r"crs::(?:\(anonymous namespace\)::)?(\w+)", name) or re.search(r"(\w+)_kernel", name) or re.search(r"(\w+)", name)
    return m.group(1)[:28] if m else name[:28]


def main():
    d = sys.argv[1]
    nb = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)[-1]
    rows = []
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], short(r["Kernel_Name"])))
    rows.sort()
    scans = [r for r in rows if r[3].startswith("scan_")]
    if not scans:
        print("no scan kernels in", f)
        return
    # the timed region of bench.py = the longest run of scans whose start-to-start distance stays under 3 x the median one
    st = [r[0] for r in scans]
    df = sorted(b - a for a, b in zip(st, st[1:]))
    lim = 3 * df[len(df) // 2]
    best, cur = (0, 0), 0
    for i in range(1, len(st)):
        if st[i] - st[i - 1] > lim:
            cur = i
        if i - cur > best[1] - best[0]:
            best = (cur, i)
    a, b = best
    q1, q3 = a + (b - a) // 4, a + 3 * (b - a) // 4
    lo, hi = st[q1], st[q3]
    win = [r for r in rows if lo <= r[0] < hi]
    byq = collections.defaultdict(list)
    for r in win:
        byq[r[2]].append(r)
    print(f"trace {f}\nwindow {1e-6 * (hi - lo):.2f} ms, {len(win)} kernels on {len(byq)} queues")
    scan_q = collections.Counter(r[2] for r in win if r[3].startswith("scan_")).most_common(1)[0][0]
    for q, rs in sorted(byq.items()):
        busy = sum(e - s for s, e, _, _ in rs)
        fam = collections.Counter(n for _, _, _, n in rs)
        print(f"  queue {q}{' (scan)' if q == scan_q else ''}: {len(rs)} kernels, busy {busy / (hi - lo):.3f}  " + ", ".join(f"{n} x{c}" for n, c in fam.most_common(6)))
    rs = byq[scan_q]
    idx = [i for i, r in enumerate(rs) if r[3].startswith("scan_")]
    if len(idx) > nb + 1:
        a = idx[len(idx) // 2]
        b = idx[len(idx) // 2 + nb]
        base = rs[a][0]
        print(f"\n{nb} consecutive batches on the scan queue (us):  start   dur   gap-before")
        for i in range(a, b + 1):
            s, e, _, n = rs[i]
            gap = s - rs[i - 1][1] if i else 0
            print(f"    {n:30s} {1e-3 * (s - base):9.1f} {1e-3 * (e - s):8.1f} {1e-3 * gap:8.1f}")
        per = [(rs[idx[j + 1]][0] - rs[idx[j]][0]) for j in range(len(idx) - 1)]
        kern, gaps = [], []
        for j in range(len(idx) - 1):
            seg = rs[idx[j]:idx[j + 1]]
            kern.append(sum(e - s for s, e, _, _ in seg))
            gaps.append((rs[idx[j + 1]][0] - rs[idx[j]][0]) - kern[-1])
        n = len(per)
        print(f"\nscan-to-scan period {1e-3 * sum(per) / n:.1f} us = kernels {1e-3 * sum(kern) / n:.1f} + gaps {1e-3 * sum(gaps) / n:.1f}  ({n} batches)")
        # which gap is it?  mean gap before each kernel family on the scan queue
        g = collections.defaultdict(list)
        for i in range(1, len(rs)):
            g[rs[i][3]].append(rs[i][0] - rs[i - 1][1])
        print("mean gap before:  " + ",  ".join(f"{n} {1e-3 * sum(v) / len(v):.1f}" for n, v in g.items()))
    for q, rs in sorted(byq.items()):
        if q == scan_q:
            continue
        starts = [i for i, r in enumerate(rs) if r[3].startswith("embed")]
        spans, kt = [], []
        for j in range(len(starts) - 1):
            seg = rs[starts[j]:starts[j + 1]]
            pool = [i for i, r in enumerate(seg) if r[3].startswith("pool")]
            if pool:
                seg = seg[:pool[0] + 1]
            spans.append(seg[-1][1] - seg[0][0])
            kt.append(sum(e - s for s, e, _, _ in seg))
        if spans:
            print(f"  queue {q}: encoder forward span {1e-3 * sum(spans) / len(spans):.1f} us, kernel time {1e-3 * sum(kt) / len(kt):.1f} us, {len(spans)} forwards")


if __name__ == "__main__":
    main()
