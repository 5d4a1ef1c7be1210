"""ORACLE -- test infrastructure only.

CPU restatements of the reference's algorithms for the hot path, each
function citing the reference file:line it follows.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this package, and only as the checker -- never as the
thing measured or shipped.  The product (tobac_flow_amd) never imports it.
"""
