"""rocprofv3 `--pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace` CSVs -> per-kernel matrix-core utilisation.

    python tools/mfma_busy.py <dir with *_counter_collection.csv and *_kernel_trace.csv> out.csv

mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (duration_ns * 2.4 cycles/ns * 1024 SIMDs): the fraction of the chip's matrix-pipe
cycles a launch kept busy (MI355X_MICROARCH.md: the counter counts cycles, 32 per v_mfma_f32_32x32x16_bf16)."""
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
    agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
    seen = set()
    with open(cc) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != "SQ_VALU_MFMA_BUSY_CYCLES":
                continue
            key = r["Dispatch_Id"]
            if key in seen:
                continue
            seen.add(key)
            # full kernel name with its template arguments: demangled names start with "void sf::(anonymous namespace)::", so cutting at the
            # first "(" merged every kernel into one row
            a = agg[r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("\"", "").split("(sf::")[0].split("(float")[0].split("(int")[0][:120]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
            a[2] += dur.get(key, 0)
    with open(out, "w") as fo:
        fo.write("# mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (duration_ns * 2.4 cycles/ns * 1024 SIMDs), per-launch averages\n")
        fo.write("kernel,launches,SQ_VALU_MFMA_BUSY_CYCLES_per_launch,duration_ns_per_launch,mfma_busy_frac\n")
        for k, (n, c, t) in sorted(agg.items(), key=lambda kv: -kv[1][2]):
            if n and t:
                fo.write(f"\"{k}\",{n},{c / n:.0f},{t / n:.0f},{c / (t * 2.4 * 1024):.4f}\n")
    print(open(out).read()[:3000])


if __name__ == "__main__":
    main()
