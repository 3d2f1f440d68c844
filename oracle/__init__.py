"""CPU oracle for the Infernos speech hot path -- TEST INFRASTRUCTURE, not product.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package.  infernos_amd/ never does (tests/test_abi.py::test_no_oracle_import_in_product enforces it).
"""
