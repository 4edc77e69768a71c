import json, sys
sys.path.insert(0, ".")
from tools_measure import config
res = {}
res["k7"] = config(7, "varKode", 1000, 256, 0)
res["k9_cgr_100"] = config(9, "cgr", 100, 100, 0, reps=2)
res["k8_cgr_100"] = config(8, "cgr", 100, 100, 0, reps=2)
res["k9_cgr_100_skew"] = config(9, "cgr", 100, 100, 1, reps=2)
print(json.dumps(res))
