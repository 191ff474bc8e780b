"""ctypes view of the C host layer built as a shared library (apemost_amd/host, `make SHARED=1`):
the reference's `mcmc` struct (src/mcmc_struct.h:30-106) and the handful of API functions the
tests call.  Test infrastructure only."""
import ctypes as C
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "apemost_amd", "host")
STRICT = "-std=c99 -fopenmp -Wall -Werror -Wextra -ansi -pedantic"   # the reference Makefile's CFLAGS

_dp = C.POINTER(C.c_double)


class GslVector(C.Structure):
    _fields_ = [("size", C.c_size_t), ("stride", C.c_size_t), ("data", _dp), ("block", C.c_void_p),
                ("owner", C.c_int)]


class GslMatrix(C.Structure):
    _fields_ = [("size1", C.c_size_t), ("size2", C.c_size_t), ("tda", C.c_size_t), ("data", _dp),
                ("block", C.c_void_p), ("owner", C.c_int)]


class Mcmc(C.Structure):
    _fields_ = [("n_par", C.c_uint), ("accept", C.c_ulong), ("reject", C.c_ulong), ("prob", C.c_double),
                ("prior", C.c_double), ("prob_best", C.c_double), ("random", C.c_void_p),
                ("params", C.POINTER(GslVector)), ("params_best", C.POINTER(GslVector)), ("files", C.c_void_p),
                ("params_descr", C.POINTER(C.c_char_p)), ("params_accepts", C.POINTER(C.c_ulong)),
                ("params_rejects", C.POINTER(C.c_ulong)), ("params_step", C.POINTER(GslVector)),
                ("params_min", C.POINTER(GslVector)), ("params_max", C.POINTER(GslVector)),
                ("data", C.POINTER(GslMatrix)), ("n_iter", C.c_ulong), ("additional_data", C.c_void_p)]


class Tempering(C.Structure):   # parallel_tempering_mcmc, src/parallel_tempering_beta.h:65-76
    _fields_ = [("beta", C.c_double), ("swapcount", C.c_ulong)]


CALLBACK = C.CFUNCTYPE(C.c_double, C.POINTER(Mcmc), C.c_void_p)


def make(out, app=None, main=None, ccflags="", strict=STRICT, shared=False):
    cmd = ["make", "-s", "-C", HOST, "OUT=" + out, "STRICT=" + strict, "CCFLAGS=" + ccflags]
    if shared:
        cmd.append("SHARED=1")
    if app:
        cmd.append("APP=" + app)
    if main:
        cmd.append("MAIN=" + main)
    subprocess.check_call(cmd)
    return out


def load(path):
    """dlopen the host layer and declare the prototypes the tests use"""
    L = C.CDLL(path, mode=C.RTLD_GLOBAL)
    mp = C.POINTER(Mcmc)
    L.mcmc_load.restype = mp
    L.mcmc_load.argtypes = [C.c_char_p, C.c_char_p]
    L.mcmc_load_params.restype = mp
    L.mcmc_load_params.argtypes = [C.c_char_p]
    L.mcmc_reuse_data.argtypes = [mp, mp]
    L.mcmc_free.restype = mp
    L.mcmc_free.argtypes = [mp]
    L.calc_model.argtypes = [mp, C.c_void_p]
    L.set_beta.argtypes = [mp, C.c_double]
    L.get_beta.restype = C.c_double
    L.get_beta.argtypes = [mp]
    L.markov_chain_step.argtypes = [mp]
    L.markov_chain_step_for.argtypes = [mp, C.c_uint]
    L.burn_in.argtypes = [mp, C.c_uint]
    L.mcmc_check_best.argtypes = [mp]
    L.markov_chain_calibrate.argtypes = [mp, C.c_uint, C.c_double, C.c_double, C.c_uint, C.c_double, C.c_double]
    L.assess_acceptance_rate.restype = C.c_uint
    L.assess_acceptance_rate.argtypes = [mp, C.c_uint, C.c_double, C.c_double, C.c_double, _dp, _dp]
    L.tempering_interaction.argtypes = [C.POINTER(mp), C.c_uint, C.c_ulong]
    L.get_chain_beta.restype = C.c_double
    L.get_chain_beta.argtypes = [C.c_uint, C.c_uint, C.c_double]
    L.calc_beta_0.restype = C.c_double
    L.calc_beta_0.argtypes = [mp, C.POINTER(GslVector)]
    L.gsl_vector_alloc.restype = C.POINTER(GslVector)
    L.gsl_vector_alloc.argtypes = [C.c_size_t]
    L.gsl_vector_free.argtypes = [C.POINTER(GslVector)]
    L.apemost_chain_place.argtypes = [mp, C.c_ulong]
    return L


def vec(v):
    return [v.contents.data[i * v.contents.stride] for i in range(v.contents.size)]


def set_vec(v, values):
    for i, x in enumerate(values):
        v.contents.data[i * v.contents.stride] = x


def attach_tempering(L, m, beta=1.0):
    """the reference's pattern (apps/eval_main.c:50, benchmark_main.c:59): the application mallocs
    sizeof(parallel_tempering_mcmc) itself and calls set_beta; returns the keep-alive object"""
    t = Tempering()
    m.contents.additional_data = C.cast(C.pointer(t), C.c_void_p)
    L.set_beta(m, beta)
    return t
