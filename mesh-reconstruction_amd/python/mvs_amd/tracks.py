"""Minimal reader of the OpenCV-FileStorage YAML dialect the tracks files use (harness for tests and bench.py; the
product reader is the C++ Configuration in mesh-reconstruction_amd/host/)."""
import os

import numpy as np
import yaml

DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "..", "tests", "data", "tracks")


class _Loader(yaml.SafeLoader):
    pass


def _matrix(loader, node):
    m = loader.construct_mapping(node, deep=True)
    return np.array(m["data"], np.float32).reshape(m["rows"], m["cols"])


_Loader.add_constructor("tag:yaml.org,2002:opencv-matrix", _matrix)


def load(name):
    text = open(os.path.join(DATA, name)).read()
    if text.startswith("%YAML:1.0"):  # OpenCV's header is not a valid YAML directive
        text = text[len("%YAML:1.0"):]
    doc = yaml.load(text, Loader=_Loader)
    cams = sorted(doc["camera"], key=lambda c: c["frame"])
    return {
        "width": int(doc["clip"]["width"]), "height": int(doc["clip"]["height"]),
        "cameras": [c["projection"] for c in cams], "near": [c["near"] for c in cams], "far": [c["far"] for c in cams],
        "bundles": np.stack([t["bundle"].reshape(4) for t in doc["tracks"]]),
    }
