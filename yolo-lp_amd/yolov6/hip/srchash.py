"""Hash of the kernel sources of libyololp_hip.so: ties a PMC traffic measurement (profiles/*_pmc_traffic.json, taken
in a separate rocprofv3 --pmc pass) to the code state it was taken on, so that bench.py never reports stale counters."""
import glob
import hashlib
import os

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'csrc')


def source_hash():
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(CSRC, '*.hip')) + glob.glob(os.path.join(CSRC, '*.inc')) + glob.glob(os.path.join(CSRC, '*.h'))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, 'rb').read())
    return h.hexdigest()[:16]
