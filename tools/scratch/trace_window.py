import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
i = int(sys.argv[2])
t0 = int(rows[i]["Start_Timestamp"])
print("%-52s %10s %10s" % ("kernel", "start us", "dur us"))
for r in rows[i:i + int(sys.argv[3])]:
    print("%-52s %10.2f %10.2f" % (r["Kernel_Name"][:52], (int(r["Start_Timestamp"]) - t0) / 1e3,
                                   (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
