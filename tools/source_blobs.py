"""git blob ids of the kernel sources a PMC summary was measured on (computed from the file contents: the GPU box has no .git).
tools/pmc_traffic.py and tools/pmc_mfma.py write them into profiles/*.json; bench.py compares them with the tree it runs from and marks the cited
figures `stale` when they differ."""
import hashlib, os

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SOURCES = ("noisediff_amd/csrc/conv3x3_wino4.hip", "noisediff_amd/csrc/pointwise.hip", "noisediff_amd/csrc/pwchain.hip", "noisediff_amd/csrc/norm.hip")


def blob_id(path):
    data = open(path, "rb").read()
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


def source_blobs(repo=REPO):
    return {p: blob_id(os.path.join(repo, p)) for p in SOURCES if os.path.exists(os.path.join(repo, p))}
