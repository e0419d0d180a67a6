"""rocprofv3 `--pmc <counters> --kernel-trace` CSVs -> per-kernel per-launch averages of every collected counter.

    python tools/pmc_by_kernel.py <dir with *_counter_collection.csv and *_kernel_trace.csv> out.csv

One line per kernel: launches, mean duration, each counter's mean per launch; for the SQ wave-state counters also the share of
SQ_WAVE_CYCLES (MI355X_MICROARCH.md: WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES; WAIT_ANY = parked on s_waitcnt / barrier)."""
import collections
import csv
import glob
import os
import sys


def main():
    d, out = sys.argv[1:3]
    cc = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    dur = {}
    with open(kt) as f:
        for r in csv.DictReader(f):
            dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    names = []
    agg = collections.defaultdict(lambda: {"n": set(), "t": 0.0, "c": collections.defaultdict(float)})
    seen = set()
    with open(cc) as f:
        for r in csv.DictReader(f):
            cn, key = r["Counter_Name"], r["Dispatch_Id"]
            if (cn, key) in seen:
                continue
            seen.add((cn, key))
            if cn not in names:
                names.append(cn)
            a = agg[r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("\"", "")[:110]]
            if key not in a["n"]:
                a["n"].add(key)
                a["t"] += dur.get(key, 0)
            a["c"][cn] += float(r["Counter_Value"])
    shares = [c for c in names if c != "SQ_WAVE_CYCLES" and c.startswith("SQ_")] if "SQ_WAVE_CYCLES" in names else []
    with open(out, "w") as fo:
        fo.write("# per-launch means; *_share = counter / SQ_WAVE_CYCLES\n")
        fo.write("kernel,launches,duration_ns," + ",".join(names) + "".join("," + c + "_share" for c in shares) + "\n")
        for k, a in sorted(agg.items(), key=lambda kv: -kv[1]["t"]):
            n = len(a["n"])
            if not n:
                continue
            row = [f"\"{k}\"", str(n), f"{a['t'] / n:.0f}"] + [f"{a['c'][c] / n:.0f}" for c in names]
            wc = a["c"].get("SQ_WAVE_CYCLES", 0.0)
            row += [f"{(a['c'][c] / wc if wc else 0.0):.3f}" for c in shares]
            fo.write(",".join(row) + "\n")
    print(open(out).read()[:4000])


if __name__ == "__main__":
    main()
