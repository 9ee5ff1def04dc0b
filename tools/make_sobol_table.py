#!/usr/bin/env python3
"""Writes sobol_10k_200.bin, the file the reference's DevScene::create reads when SAMPLER_USE_SOBOL is on (src/scene.cpp:500-506):
10 000 x 200 uint32 of the unscrambled Sobol sequence (Joe-Kuo direction numbers from scipy; restir_amd/sobol.py).

    python tools/make_sobol_table.py [out=sobol_10k_200.bin]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from restir_amd import sobol

if __name__ == "__main__":
    out = sys.argv[1] if len(sys.argv) > 1 else "sobol_10k_200.bin"
    t = sobol.write_table(out)
    print(f"{out}: {t.shape[0]} x {t.shape[1]} uint32, sha256 {sobol.table_digest(t)}"
          + ("" if sobol.table_digest(t) == sobol.TABLE_SHA256 else "  (differs from the digest recorded in restir_amd/sobol.py: another scipy?)"))
