"""CPU oracle loader -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package.  The product (eepacc_mpc_casadi_matlab_amd) never does.
"""
from .loader import Oracle, build_oracle  # noqa: F401
