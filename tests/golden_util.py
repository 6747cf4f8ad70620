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


R_MOD = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001


def aggregate(scalars, bases):
    """merge repeated bases (first-appearance order, scalars summed mod r): the form in which a GWC Guard is reported"""
    order, acc = [], {}
    for s, b in zip(scalars, bases):
        if b not in acc:
            order.append(b); acc[b] = 0
        acc[b] = (acc[b] + int.from_bytes(s, "little")) % R_MOD
    return [acc[b].to_bytes(32, "little") for b in order], order
