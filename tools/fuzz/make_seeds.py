"""seed corpus of tools/fuzz/fuzz_readers.cpp: one small valid file per reader (first byte = reader selector) and the four bundled tracks
files cut down to a few cameras and tracks.  usage: make_seeds.py <corpus dir>"""
import os
import re
import sys

out = sys.argv[1]
os.makedirs(out, exist_ok=True)
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def put(name, selector, payload):
    open(os.path.join(out, name), "wb").write(bytes([selector]) + payload)


put("pgm", 0, b"P5\n4 3\n255\n" + bytes(range(12)))
put("ppm", 1, b"P6\n2 2\n255\n" + bytes(range(12)))
for tag, sel in (("420jpeg", 2), ("422", 6), ("444", 10), ("mono", 14)):
    cw, ch = {"420jpeg": (2, 2), "422": (2, 1), "444": (1, 1), "mono": (0, 0)}[tag]
    w, h = 6, 4
    chroma = 0 if not cw else 2 * ((w + cw - 1) // cw) * ((h + ch - 1) // ch)
    frame = b"FRAME\n" + bytes((7 * i) & 255 for i in range(w * h + chroma))
    put("y4m_" + tag, sel | 0x10, ("YUV4MPEG2 W%d H%d F25:1 Ip A1:1 C%s\n" % (w, h, tag)).encode() + frame * 3)
for name in sorted(os.listdir(os.path.join(ROOT, "tests", "data", "tracks"))):
    text = open(os.path.join(ROOT, "tests", "data", "tracks", name)).read()
    head, _, rest = text.partition("camera:")
    cams = re.split(r"(?m)^(?=\s*- frame:)", rest)
    body, _, tracks = cams[-1].partition("tracks:")
    cams[-1] = body
    tr = re.split(r"(?m)^(?=\s*- bundle:)", tracks)
    small = head + "camera:" + "".join(cams[:4]) + "tracks:" + "".join(tr[:4])
    put("yaml_" + name, 3, small.encode())
    put("yaml_k2_" + name, 7, small.encode())
print("seeds written to", out)
