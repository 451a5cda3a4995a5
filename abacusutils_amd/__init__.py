"""abacusutils_amd - MI355X-native hot path of abacusorg/abacusutils.

Drop-in module layout for the calls on the path:

    abacusutils_amd.hod.abacus_hod.AbacusHOD        (run_hod, compute_power, compute_xirppi, ...)
    abacusutils_amd.hod.GRAND_HOD.gen_gal_cat
    abacusutils_amd.analysis.power_spectrum.calc_power, calc_pk_from_deltak, get_k_mu_edges, ...
    abacusutils_amd.analysis.tsc.tsc_parallel, partition_parallel
    abacusutils_amd.analysis.tpcf_corrfunc.calc_xirppi_fast, calc_wp_fast, calc_multipole_fast

Host code is Python over a C ABI (include/abacus_hip.h, ctypes) to hand-written HIP kernels for gfx950.
There is no CPU fallback: without libabacus_hip.so and a GPU the compute calls raise.
"""
__version__ = '0.1.0'
