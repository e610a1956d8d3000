"""Per-stream coefficient generators for the BASELINE workloads: moved into the package (zignal_amd/workloads.py)."""
from zignal_amd.workloads import osc_chain_params  # noqa: F401
