"""Generates tests/golden/ref_etm2d.npz from the REFERENCE's own Python check of its elastoplastic tangent.

tests/Constitutive/Elastoplastic-Tangent-Matrix.py (reference tree) re-derives with numpy, for one hard-coded
2 x 2 case, the material part of the spectral stiffness density the C driver next to it
(tests/Constitutive/Elastoplastic-Tangent-Matrix.c:85-175, same numbers) evaluates.  This script runs that file
unmodified (numpy only, it prints three matrices), takes its inputs and its A_ep from the module namespace and
stores them; nothing of its text is kept.  The fixture pins the oracle's spectral stiffness density
(oracle/nlps_oracle.c::orc_stiffness_density_spectral) in tests/test_oracle.py.

    python tests/golden/make_ref_fixtures.py      (needs /root/reference; run in the build container only)
"""
import contextlib
import io
import os
import runpy

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/nl-partsol/tests/Constitutive/Elastoplastic-Tangent-Matrix.py"


def main():
    with contextlib.redirect_stdout(io.StringIO()):
        ns = runpy.run_path(REF)
    np.savez(os.path.join(HERE, "ref_etm2d.npz"),
             dN_alpha=np.asarray(ns["dN_alpha"], dtype=np.float64), dN_beta=np.asarray(ns["dN_beta"], dtype=np.float64),
             tau=np.asarray(ns["tau"], dtype=np.float64), D_phi=np.asarray(ns["D_phi"], dtype=np.float64),
             b_e=np.asarray(ns["b_e"], dtype=np.float64), a_ep=np.asarray(ns["a_ep"], dtype=np.float64),
             u=np.asarray(ns["u"], dtype=np.float64), v=np.asarray(ns["v"], dtype=np.float64),
             A_ep=np.asarray(ns["A_ep"], dtype=np.float64))
    print("ref_etm2d.npz: A_ep =\n", ns["A_ep"])


if __name__ == "__main__":
    main()
