"""Do kernels of different contexts / HIP streams overlap?  Parses a rocprofv3 --kernel-trace csv."""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_sos_os" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Stream_Id", r.get("Queue_Id", "?"))))
rows.sort()
ov = sum(1 for a, b in zip(rows[:-1], rows[1:]) if b[0] < a[1])
print("k_sos_os launches", len(rows), "overlapping successors", ov, "queues/streams", sorted(set(r[2] for r in rows))[:10])
print([(r[0] - rows[0][0], r[1] - rows[0][0], r[2]) for r in rows[-8:]])
