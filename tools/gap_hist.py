#!/usr/bin/env python3
"""Where do the gaps of the step's dependent chain come from?  

    gap_hist.py <kernel_trace.csv> [<hip_api_trace.csv>]

Takes ONE steady-state step of a `rocprofv3 --kernel-trace [--hip-runtime-trace]` run of bench.py (the span between the last two
add_noise launches), walks the queue that carries the most launches (the student's dependent chain) and prints
  * the distribution of the gaps between consecutive kernels of that queue,
  * the gaps grouped by (kind of predecessor -> kind of successor),
  * with the HIP API trace: the gaps split by what the host enqueued between the two launches -- nothing, an event record / stream
    wait (a fork to or a join from a side stream), or launches on other streams -- joined through the correlation ids,
  * how many other-queue kernels were running while each gap lasted (contention for the CUs by the side streams).
"""
import collections
import csv
import sys


def kind(name):
    n = name.split("(")[0]
    for k in ("norm_partial", "norm_apply_head", "norm_apply", "norm_bwd_apply", "norm_finalize", "norm_fused_fwd", "norm_fused_bwd",
              "norm_head", "norm_sum", "splitk_finish", "conv_k3_tile", "conv_k3_lds", "conv_k3_p32", "conv_k3_p16", "conv_k3_c1",
              "conv_gemm", "conv_smallk3", "wgrad", "reduce_partials", "first_block", "seg_losses", "fecl", "l2norm", "sgd_ema", "sumsq",
              "pack_batch", "add_noise", "colsum", "trilinear", "cast", "level_block"):
        if k in n:
            return k
    return n[-24:]


def pct(xs, p):
    xs = sorted(xs)
    return xs[min(len(xs) - 1, int(p * len(xs)))] if xs else 0.0


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    starts = [i for i, r in enumerate(rows) if "add_noise" in r["Kernel_Name"]]
    a, b = starts[-2], starts[-1]
    seg = rows[a:b]
    qkey = "Queue_Id" if "Queue_Id" in seg[0] else "Stream_Id"
    byq = collections.defaultdict(list)
    for r in seg:
        byq[r[qkey]].append(r)
    main_q = max(byq, key=lambda q: len(byq[q]))
    chain = byq[main_q]
    others = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in seg if r[qkey] != main_q]
    wall = (int(rows[b]["Start_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e3
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in chain) / 1e3
    print(f"step wall {wall:.1f} us; chain queue {main_q}: {len(chain)} launches, {busy:.1f} us of kernels; {len(seg) - len(chain)} launches on "
          f"{len(byq) - 1} other queues")

    api_between = None
    if len(sys.argv) > 2:
        api = list(csv.DictReader(open(sys.argv[2])))
        api.sort(key=lambda r: int(r["Start_Timestamp"]))
        pos = {r["Correlation_Id"]: i for i, r in enumerate(api)}
        names = [r["Function"] for r in api]
        chain_corr = {r["Correlation_Id"] for r in chain}

        def between(r0, r1):
            i0, i1 = pos.get(r0["Correlation_Id"]), pos.get(r1["Correlation_Id"])
            if i0 is None or i1 is None:
                return "unknown"
            mid = names[i0 + 1:i1]
            ev = sum(1 for f in mid if f in ("hipEventRecord", "hipStreamWaitEvent"))
            ln = sum(1 for f in mid if "Launch" in f)
            if ev and ln:
                return "fork: event ops + launches on another stream"
            if ev:
                return "event ops only (join / mark)"
            if ln:
                return "launches on another stream only"
            return "nothing between the two launches"
        api_between = between
        del chain_corr

    gaps, by_pair, by_api, by_load = [], collections.defaultdict(list), collections.defaultdict(list), collections.defaultdict(list)
    prev = None
    for r in chain:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if prev is not None:
            ps, pe = int(prev["Start_Timestamp"]), int(prev["End_Timestamp"])
            g = (s - pe) / 1e3
            gaps.append(g)
            by_pair[(kind(prev["Kernel_Name"]), kind(r["Kernel_Name"]))].append(g)
            if api_between is not None:
                by_api[api_between(prev, r)].append(g)
            running = sum(1 for (os_, oe) in others if os_ < s and oe > pe)      # other-queue kernels alive at some point of the gap
            by_load["%d other-queue kernel(s) alive during the gap" % min(running, 3) + ("+" if running >= 3 else "")].append(g)
        prev = r
    pos_g = [g for g in gaps if g > 0]
    print(f"gaps: {len(gaps)}, sum {sum(pos_g):.1f} us, median {pct(pos_g, .5):.2f}, p10 {pct(pos_g, .1):.2f}, p90 {pct(pos_g, .9):.2f}, max {max(gaps):.1f}")
    edges = [0, 1, 2, 3, 4, 6, 8, 12, 20, 50, 1e9]
    print("histogram (us):  " + "  ".join(f"[{lo:g},{hi:g}): {sum(1 for g in gaps if lo <= g < hi)}" for lo, hi in zip(edges[:-1], edges[1:])))

    def table(title, d):
        print(f"\n{title}")
        for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
            k = " -> ".join(k) if isinstance(k, tuple) else k
            print(f"  {sum(v):8.1f} us  n {len(v):3d}  median {pct(v, .5):6.2f}  p90 {pct(v, .9):6.2f}   {k}")
    if by_api:
        table("by what the host enqueued between the two launches:", by_api)
    table("by concurrency:", by_load)
    table("by (predecessor -> successor), top 25:", dict(sorted(by_pair.items(), key=lambda kv: -sum(kv[1]))[:25]))


if __name__ == "__main__":
    main()
