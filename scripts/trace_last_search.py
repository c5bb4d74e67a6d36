"""Timeline of the last search in a rocprofv3 kernel trace (csv): python scripts/trace_last_search.py <dir> [marker kernel]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
marker = sys.argv[2] if len(sys.argv) > 2 else "fill_f32_kernel"
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]][-1]
t0 = int(rows[max(0, idx - 14)]["Start_Timestamp"])
for r in rows[max(0, idx - 14):]:
    print("%9.1f us  %8.1f us  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Kernel_Name"][:100]))
