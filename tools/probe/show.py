"""print the headline fields of bench.py JSON lines:  python tools/probe/show.py a.json [b.json ...]"""
import json, sys
for f in sys.argv[1:]:
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, "value", d["value"], d.get("train_mode"), "eager", d.get("train_images_per_sec_eager"), "graph", d.get("train_images_per_sec_graph"), "ms", d["ms_per_step"],
          "infer", d.get("infer_images_per_sec"), d.get("infer_images_per_sec_eager"), "frac", d["roofline"]["frac"])
