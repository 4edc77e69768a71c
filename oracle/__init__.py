"""CPU oracle for the varKoder `image` hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  The product (varkoder_amd) never does.
"""
