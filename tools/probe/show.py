"""print the headline fields of bench.py JSON lines:  python tools/probe/show.py a.json [b.json ...]"""
import json, sys
for f in sys.argv[1:]:
    d = json.load(open(f))
    print(f, d["value"], d["ms_per_step"], d.get("infer_images_per_sec"), d.get("infer_images_per_sec_eager"), d["roofline"]["frac"])
