from . import BIN_DIR

print(BIN_DIR)
