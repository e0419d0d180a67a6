"""rocprofv3 PMC passes -> per-kernel HBM-side traffic per launch (profiles/*_pmc_traffic.json).

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d F -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 --no-graph
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d W -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 --no-graph
    python tools/pmc_traffic.py F/.../*_counter_collection.csv W/.../*_counter_collection.csv profiles/r1_k_pmc_traffic.json

Units and corrections as /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section) prescribes: the counters are in
KiB; on gfx950 FETCH_SIZE tallies the 128-byte requests of wide (16 B / lane) coalesced reads at 64 bytes, so it is DOUBLED;
WRITE_SIZE is exact for 16-byte streaming stores.  Separate passes (FETCH_SIZE and WRITE_SIZE do not fit one pass).
Infinity-Cache hits are included in both, i.e. this is traffic at the L2's memory side, an upper bound of HBM bytes.
"""
import collections
import csv
import json
import re
import sys

LABELS = [  # (regex on the kernel name, bench.py label prefix)
    (r"conv_cb_kernel", "conv_cb"), (r"cb_reduce_gn_kernel", "cb_reduce_gn"), (r"cb_reduce_ln_kernel", "cb_reduce_ln"),
    (r"conv_gemm_rs_kernel", "conv_gemm_rs"), (r"conv_gemm_fast_kernel", "conv_gemm_fast"), (r"conv_gemm_wp_kernel", "conv_gemm_wp"), (r"conv_gemm_v2_kernel", "conv_gemm_v2"),
    (r"conv_gemm_mt_kernel", "conv_gemm_mt"), (r"thin_tail_kernel", "conv_thin"), (r"attention_ksplit_kernel", "attention"),
    (r"conv_gemm_sk_kernel", "conv_gemm_sk"), (r"conv_gemm_kernel", "conv_gemm"), (r"conv_thin_kernel", "conv_thin"),
    (r"conv_direct_kernel", "conv_direct"), (r"gn_silu_kernel", "gn_silu"), (r"gn_stats_kernel", "gn_stats"),
    (r"ln_modulate_kernel", "ln_modulate"), (r"attention_mfma_kernel|attention_kernel", "attention"),
]


def label_of(name):
    for rx, lab in LABELS:
        if re.search(rx, name):
            return lab
    return None


def collect(path, counter):
    d = collections.defaultdict(lambda: [0, 0.0])
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            lab = label_of(r["Kernel_Name"])
            if lab:
                d[lab][0] += 1
                d[lab][1] += float(r["Counter_Value"])
    return d


def main():
    fetch, write, out = sys.argv[1:4]
    f, w = collect(fetch, "FETCH_SIZE"), collect(write, "WRITE_SIZE")
    res = {}
    for lab in sorted(set(f) | set(w)):
        nf, vf = f.get(lab, [0, 0.0])
        nw, vw = w.get(lab, [0, 0.0])
        fk = vf / nf if nf else 0.0
        wk = vw / nw if nw else 0.0
        res[lab] = {"launches_sampled": nf, "fetch_kib_raw_per_launch": round(fk, 2), "write_kib_per_launch": round(wk, 2),
                    "traffic_bytes_per_launch": round((2.0 * fk + wk) * 1024.0)}
    meta = {"_note": "FETCH_SIZE doubled (gfx950 wide-read correction), WRITE_SIZE as is, KiB -> bytes; per launch, 2 clip-parallel "
                     "branches (batch 4 per launch); includes Infinity-Cache hits", "kernels": res}
    with open(out, "w") as fo:
        json.dump(meta, fo, indent=1)
    for k, v in res.items():
        print(k, v)


if __name__ == "__main__":
    main()
