import glob
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))


def cases():
    out = []
    for p in sorted(glob.glob(os.path.join(HERE, "golden", "*.json"))):
        with open(p) as f:
            out.append(json.load(f))
    return out


def h(x):
    return bytes.fromhex(x)


def instances_of(entry):
    return [[h(v) for v in col] for col in entry["instances"]]
