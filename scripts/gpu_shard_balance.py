#!/usr/bin/env python3
"""Load balance over the k-mer-prefix shards of ONE build at N = 2, 4, 8, measured shard by shard on a box with one GPU.

  python scripts/gpu_shard_balance.py --workload real10x600M [--shards 2,4,8] [--modes rescan,exchange] [--range-cap KEYS]

All N shards live on GPU 0 (debwt_multi_create with a repeated ordinal) and take turns between the barriers of the build
(debwt_multi_set_serial): one shard on the GPU at a time, device drained around every step, so a step's wall time is that
shard's own.  Per (N, key mode): one warm-up build, one measured build, the device inverse BWT of the concatenated result,
then per shard the step times (debwt_multi_get_shard_report) grouped into stages, keys / blue rows / blocks / large
blocks, and bytes in and out of every exchange; per stage max / mean over the shards; the replicated work (steps whose time
does not shrink with N); and the PROJECTED critical path of a real N-GPU node = per-stage max over the shards + the
exchanges at an assumed link rate (labelled as a projection: no multi-GPU box was available).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

STAGES = {
    "census+plan": ("shard_histogram", "shard_plan"),
    "keys+sort": ("kmer_sort_rle", "shard_partition_keys", "shard_sort_range", "shard_sort_end"),
    "classify": ("shard_classify_local", "shard_facts_export", "shard_classify_global"),
    "sp": ("shard_sp_flags", "shard_sp_emit", "shard_sp_import"),
    "blue": ("shard_blue_route", "shard_blue_place", "blue_sort"),
    "assemble+concat": ("bwt_assemble", "shard_export", "concat_rows"),
}
EXCH = ("exchange_keys", "exchange_facts", "exchange_sp", "exchange_blue", "exchange_rows")
# steps whose work is the same on every GPU whatever N (every GPU holds the whole text, red table and SP code)
REPLICATED = ("shard_classify_global", "shard_sp_import")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="real10x600M")
    ap.add_argument("--shards", default="2,4,8")
    ap.add_argument("--modes", default="rescan,exchange")
    ap.add_argument("--k", type=int, default=32)
    ap.add_argument("--range-cap", type=int, default=0, help="key instances per key range of every shard (0: the library's plan)")
    ap.add_argument("--link-gbs", type=float, default=48.0, help="assumed sustained GB/s per xGMI link and direction (projection only)")
    ap.add_argument("--tune", type=int, default=0)
    ap.add_argument("--splitters", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--json", default=None, help="also write every report as JSON lines to this file")
    args = ap.parse_args()
    from debwt_amd import api, _lib
    from debwt_amd import synth_native as SN
    t0 = time.perf_counter()
    syn = SN.Synth.named(args.workload)
    n, nrec = syn.n, syn.nrec
    sep = syn.sep()
    text = SN.PinnedArray(syn.nwords)
    syn.words_into(text.ptr)
    print(f"# {args.workload}: n = {n}, {nrec} records, k = {args.k}; text generated in {time.perf_counter() - t0:.1f} s")
    # the single-GPU build of the same text, for scale
    d = api.DeBWT(k=args.k, device=0, tune=args.tune)
    d.load_packed(text.a, n, sep)
    d.build(); d.build()
    st1 = d.stats()
    print(f"# one GPU, unsharded: {st1['ms_total']:.1f} ms (sort {st1['ms_sort']:.1f}, classify {st1['ms_classify']:.1f}, sp {st1['ms_sp']:.1f}, "
          f"blue {st1['ms_blue']:.1f}, assemble {st1['ms_assemble']:.1f}); blue rows {st1['blue_capacity']}, blocks {st1['blue_bound_num']}, "
          f"large blocks {st1['blue_large_blocks']}")
    d.close()
    jf = open(args.json, "a") if args.json else None
    for N in [int(x) for x in args.shards.split(",")]:
        for mode in args.modes.split(","):
            m = api.MultiDeBWT([0] * N, k=args.k, tune=args.tune)
            try:
                m.set_serial(True)
                m.set_key_mode(mode)
                if args.range_cap:
                    for r in range(N):
                        rc = m._L.debwt_set_range_cap(m._shard_ctx(r), args.range_cap)
                        assert rc == 0
                m.load_packed(text.a, n, sep)
                m.build()
                m.build()
                ok = m.verify_device()["ok"]
                reps = [m.shard_report(r) for r in range(N)]
                ms, _ = m.stats()
            except api.DebwtError as e:
                print(f"\n## N = {N}, keys: {mode}: {e} -- the {N} shards of this collection do not fit next to each other in ONE GPU's HBM "
                      f"(on a node every shard has a GPU of its own)")
                m.close()
                continue
            m.close()
            if jf:
                jf.write(json.dumps({"workload": args.workload, "n": n, "N": N, "mode": mode, "inverse_bwt_ok": bool(ok), "shards": reps}) + "\n")
                jf.flush()
            print(f"\n## N = {N}, keys: {mode}, key rounds {ms['rounds']}, inverse BWT of the concatenated result {'ok' if ok else 'FAILED'}")
            print("shard  bins        keys(M) ranges  blue_rows(M) blocks(k) large   " + "  ".join(f"{s_:>15}" for s_ in STAGES) + "    total   exch_in(MB) exch_out(MB)")
            tot = {s_: [] for s_ in STAGES}
            totals = []
            for r in reps:
                per = {s_: sum(r["ms"].get(x, 0.0) for x in steps) for s_, steps in STAGES.items()}
                for s_ in STAGES:
                    tot[s_].append(per[s_])
                t_all = sum(per.values())
                totals.append(t_all)
                print(f"{r['shard']:>5}  {r['bins'][0]:>4}-{r['bins'][1]:<4}  {r['keys'] / 1e6:>8.1f} {r['key_ranges']:>6}  {r['blue_rows'] / 1e6:>11.1f} "
                      f"{r['blocks'] / 1e3:>9.1f} {r['ctx']['blue_large_blocks']:>5}   " + "  ".join(f"{per[s_]:>15.1f}" for s_ in STAGES) +
                      f"  {t_all:>7.1f}   {sum(r['bytes_in'].values()) / 1e6:>10.1f} {sum(r['bytes_out'].values()) / 1e6:>11.1f}")
            print("max/mean" + " " * 57 + "  ".join(f"{(max(v) / max(np.mean(v), 1e-9)):>15.2f}" for v in tot.values()) + f"  {max(totals) / np.mean(totals):>7.2f}")
            print("max (ms)" + " " * 57 + "  ".join(f"{max(v):>15.1f}" for v in tot.values()) + f"  {max(totals):>7.1f}")
            repl = {x: float(np.mean([r["ms"].get(x, 0.0) for r in reps])) for x in REPLICATED}
            # exchanges: what the busiest shard receives or sends, over (N - 1) links in parallel at the assumed rate
            xin = {x: max(r["bytes_in"][x] for r in reps) for x in ("keys", "facts", "sp", "blue", "rows")}
            xout = {x: max(r["bytes_out"][x] for r in reps) for x in ("keys", "facts", "sp", "blue", "rows")}
            xms = {x: max(xin[x], xout[x]) / max(N - 1, 1) / (args.link_gbs * 1e9) * 1e3 for x in xin}
            xms["rows"] = xin["rows"] / (args.link_gbs * 1e9) * 1e3 / max(N - 1, 1)          # a gather into shard 0 over its N - 1 links
            crit = sum(max(v) for v in tot.values())
            print(f"replicated work per GPU (ms, mean over shards): " + ", ".join(f"{k_} {v:.1f}" for k_, v in repl.items()))
            print(f"exchange bytes of the busiest shard (MB in / out): " + ", ".join(f"{x} {xin[x] / 1e6:.0f}/{xout[x] / 1e6:.0f}" for x in xin))
            print(f"PROJECTION for a real {N}-GPU node: sum over stages of the slowest shard {crit:.1f} ms + exchanges at an ASSUMED "
                  f"{args.link_gbs:.0f} GB/s per link " + " + ".join(f"{x} {v:.1f}" for x, v in xms.items()) + f" = {crit + sum(xms.values()):.1f} ms "
                  f"-> {n / (crit + sum(xms.values())) / 1e6:.2f} Gbp/s (one GPU unsharded: {st1['ms_total']:.1f} ms = {n / st1['ms_total'] / 1e6:.2f} Gbp/s)")
            sys.stdout.flush()
    text.free()


if __name__ == "__main__":
    main()
