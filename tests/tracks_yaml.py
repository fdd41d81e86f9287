"""tracks/*.yaml reader: lives in the harness package so bench.py can use it too"""
from mvs_amd.tracks import DATA, load  # noqa: F401
