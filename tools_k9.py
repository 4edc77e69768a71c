import sys
sys.path.insert(0, ".")
from tools_measure import config
print(config(9, "cgr", 100, 100, 0, reps=2))
