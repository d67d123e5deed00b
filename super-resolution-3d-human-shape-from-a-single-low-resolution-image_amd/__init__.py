"""MI355X-native SuRS occupancy-query hot path (encoder -> fused point evaluator
-> Lewiner marching cubes) behind the reference's SuRSNet / reconstruction /
gen_mesh interface.  Import as ``surs_amd``.

Host code here is orchestration only; all arithmetic lives in the HIP library
built from ``csrc/`` and reached through the C ABI declared in
``include/surs.h``.  There is no CPU fallback: calling a compute entry point
without the built library (or without a GPU) raises.
"""
__version__ = "0.1.0"
