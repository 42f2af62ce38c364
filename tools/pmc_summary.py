import csv, glob, sys, collections
prefix, kname = sys.argv[1], sys.argv[2]
agg = collections.OrderedDict()
dur = []
for f in sorted(glob.glob(f"gpurun_out/{prefix}_*/**/*_counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if kname in r["Kernel_Name"]:
            agg.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for f in sorted(glob.glob(f"gpurun_out/{prefix}_SQ_WAVES/**/*_kernel_trace.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if kname in r["Kernel_Name"]:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print(f"{kname}: durations us {['%.0f' % d for d in dur]}")
w = agg.get("SQ_WAVES", [1])[-1]
for k, v in agg.items():
    print(f"  {k:24s} {v[-1]:14.4g}   per wave {v[-1] / w:12.1f}")
