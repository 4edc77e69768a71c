import sys
sys.path.insert(0, ".")
from tools_measure import config
print(config(9, "cgr", 64, 64, 0, reps=1))
