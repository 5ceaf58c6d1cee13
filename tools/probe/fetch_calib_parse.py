"""prints FETCH_SIZE (MB) per launch of tools/probe/fetch_calib.cpp from rocprofv3 --pmc output dirs:  python fetch_calib_parse.py <dir> ..."""
import csv, glob, sys
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r["Counter_Name"] == "FETCH_SIZE" and "gather_kernel" in r["Kernel_Name"]]
        print(d.rstrip("/").split("/")[-1], "FETCH_SIZE per launch, counter unit KiB -> MB:", [round(x * 1024 / 1e6, 1) for x in v], "(838.9 MB read)")
