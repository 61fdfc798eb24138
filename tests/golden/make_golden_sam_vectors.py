"""Turns the SAM / BAM / BAI files that the reference's vendored htslib keeps for ITS OWN tests
(/root/reference/deepmutect/htslib/test/) into tests/golden/sam_vectors.npz: the input bytes of every file (data, not source)
and, for the SAM files, the field tuples an independent spec-based parser (tests/sam_spec.py) reads from the text.
Run in the build container only (the reference tree does not travel):  python tests/golden/make_golden_sam_vectors.py"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import sam_spec  # noqa: E402

SRC = "/root/reference/deepmutect/htslib/test"
SAMS = ["auxf#values.sam", "ce#5b.sam", "ce#supp.sam", "ce#unmap.sam", "ce#unmap1.sam", "ce#unmap2.sam", "c1#clip.sam", "c1#bounds.sam",
        "ce#1.sam", "ce#2.sam", "ce#5.sam", "c1#noseq.sam", "c1#unknown.sam", "index.sam"]
BINS = ["range.bam", "range.bam.bai", "colons.bam", "colons.bam.bai", "index.bam.bai"]

out, expected = {}, {}
for name in SAMS:
    raw = open(os.path.join(SRC, name), "rb").read()
    out["sam:" + name] = np.frombuffer(raw, dtype=np.uint8)
    header, refs, recs = sam_spec.parse_sam_text(raw.decode())
    expected[name] = dict(header=header, refs=refs, records=recs)
for name in BINS:
    out["bin:" + name] = np.frombuffer(open(os.path.join(SRC, name), "rb").read(), dtype=np.uint8)
out["expected_json"] = np.frombuffer(json.dumps(expected).encode(), dtype=np.uint8)
np.savez_compressed(os.path.join(HERE, "sam_vectors.npz"), **out)
print({k: len(v) for k, v in out.items()})
print({k: len(v["records"]) for k, v in expected.items()})
