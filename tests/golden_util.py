"""Loaders for the committed golden fixtures (tests/golden/, made by oracle/gen_golden.py)."""
import base64
import json
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
STYLES = ["instant", "first", "tangent", "last", "full"]
CONFIG_DFAS = ["err", "uri", "log100", "syn256", "num3", "newyork", "aab", "dotstar_err",
               "uri_v6", "uri_user", "syn4k"]


def unb64(s):
    return base64.b64decode(s)


def load_kat():
    with open(os.path.join(GOLD, "kat_matcher.json")) as f:
        return json.load(f)


def kat_items():
    """Yields (case_name, fmt_name, blob, calls) for every KAT case x format that compiled."""
    for case in load_kat()["cases"]:
        for fmt, b in case["blobs"].items():
            if "reda" in b:
                yield case["name"], fmt, unb64(b["reda"]), case["calls"]


def load_omnibus():
    with open(os.path.join(GOLD, "omnibus.json")) as f:
        meta = json.load(f)
    blobs = np.load(os.path.join(GOLD, "omnibus_blobs.npz"))
    return meta["rows"], blobs


_DFA_CACHE = {}


def load_dfa(name):
    """The reference-compiled blob of a config DFA (big ones are stored xz-compressed)."""
    if name not in _DFA_CACHE:
        path = os.path.join(GOLD, "dfas", name + ".reda")
        if os.path.exists(path):
            with open(path, "rb") as f:
                _DFA_CACHE[name] = f.read()
        else:
            import lzma
            with open(path + ".xz", "rb") as f:
                _DFA_CACHE[name] = lzma.decompress(f.read())
    return _DFA_CACHE[name]


def load_vectors(name):
    return np.load(os.path.join(GOLD, "vectors_%s.npz" % name))


def expect_of(vec, verb, style_idx, lead):
    key = "%s_%d_%d" % (verb, style_idx, lead)
    res = vec[key + "_res"]
    if verb in ("match", "search"):
        return res, vec[key + "_start"], vec[key + "_end"]
    return res, None, None
