"""Sum one counter over the conv launches of a rocprofv3 --pmc csv: python scripts/pmc_one_conv.py <dir> <COUNTER>"""
import csv
import glob
import sys

tot, n = 0.0, 0
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == sys.argv[2] and "conv_" in r["Kernel_Name"] and "pack" not in r["Kernel_Name"]:
            tot += float(r["Counter_Value"])
            n += 1
print("%s: %d conv launches, %.1f per launch" % (sys.argv[2], n, tot / max(n, 1)))
