import csv, sys, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "step_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    v = acc[k][10:] or acc[k]
    print("%-28s avg/dispatch %14.1f   (n=%d)" % (k, sum(v) / len(v), len(v)))
