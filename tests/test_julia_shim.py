"""julia/DEMCHip.jl cannot be executed here (no Julia in the image or on the GPU box), so the one thing that can
be validated is validated: every `ccall` in the shim is parsed and its symbol, return type, argument count and
argument types are checked against the prototype in include/demcz.h, and the `DemczConfig` struct against
`demcz_config`, field by field.  (The call *sequence* is the one demc.jl_amd/sampler.py runs and the GPU tests
verify.)"""
import re
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
JL = (ROOT / "julia" / "DEMCHip.jl").read_text()
HDR = re.sub(r"/\*.*?\*/", "", (ROOT / "include" / "demcz.h").read_text(), flags=re.S)

# Julia ccall type -> the C types it may stand for
JL2C = {
    "Int64": {"int64_t"}, "Int32": {"int32_t"}, "Float64": {"double"}, "UInt64": {"uint64_t"},
    "Ptr{Float64}": {"double*", "const double*"}, "Ref{Float64}": {"double*", "const double*"},
    "Ptr{Int64}": {"int64_t*"}, "Ref{Int64}": {"int64_t*"},
    "Ptr{Int32}": {"int32_t*", "const int32_t*"}, "Ref{Int32}": {"int32_t*"},
    "Ptr{Cvoid}": {"demcz_handle*", "const demcz_handle*", "void*", "const void*"},
    "Ptr{UInt8}": {"void*", "const void*"},
    "Ref{Ptr{Cvoid}}": {"demcz_handle**", "void**"},
    "Ref{Ptr{Float64}}": {"double**"},
    "Ref{DemczConfig}": {"const demcz_config*"},
    "Cstring": {"const char*"},
}


def c_prototypes():
    protos = {}
    for ret, name, args in re.findall(r"(int32_t|const char\s*\*)\s+(demcz_\w+)\s*\(([^)]*)\)\s*;", HDR):
        types = []
        for a in [x.strip() for x in args.split(",")]:
            if a in ("void", ""):
                continue
            m = re.match(r"(.*?)(\b\w+)$", a)                      # strip the parameter name
            t = re.sub(r"\s+", " ", m.group(1)).strip().replace(" *", "*")
            types.append(t)
        protos[name] = (re.sub(r"\s+", " ", ret).replace(" *", "*"), types)
    return protos


def julia_ccalls():
    out = []
    for name, ret, args in re.findall(r"ccall\(\(:(\w+),\s*libdemcz\),\s*([\w{}]+),\s*\(([^)]*)\)", JL, flags=re.S):
        types = [t.strip() for t in args.replace("\n", " ").split(",") if t.strip()]
        out.append((name, ret, types))
    return out


def test_every_ccall_matches_its_prototype():
    protos = c_prototypes()
    calls = julia_ccalls()
    assert len(calls) >= 27
    for name, ret, types in calls:
        assert name in protos, f"{name}: not declared in include/demcz.h"
        cret, ctypes_ = protos[name]
        assert cret in JL2C[ret], f"{name}: return {ret} vs {cret}"
        assert len(types) == len(ctypes_), f"{name}: {len(types)} ccall arguments vs {len(ctypes_)} in the header"
        for i, (jt, ct) in enumerate(zip(types, ctypes_)):
            assert jt in JL2C, f"{name}: unknown Julia type {jt}"
            assert ct in JL2C[jt], f"{name}: argument {i} is {jt} in the shim, {ct} in the header"


def test_shim_binds_the_contract_functions():
    """SURVEY.md 8(b)'s list plus what the drivers need, closure mode included."""
    bound = {c[0] for c in julia_ccalls()}
    need = {"demcz_create", "demcz_destroy", "demcz_last_error", "demcz_set_state", "demcz_get_state", "demcz_run",
            "demcz_rhat", "demcz_propose", "demcz_accept_commit", "demcz_end_generation", "demcz_run_checked",
            "demcz_get_history", "demcz_get_changed", "demcz_accept_ratio", "demcz_set_rng_offset"}
    assert need <= bound, sorted(need - bound)


def test_config_struct_matches_header():
    m = re.search(r"typedef struct demcz_config \{(.*?)\} demcz_config;", HDR, flags=re.S)
    cfields = []
    for line in m.group(1).split(";"):
        line = line.strip()
        if not line:
            continue
        mm = re.match(r"(.*?)(\b\w+)$", line)
        cfields.append((mm.group(2), re.sub(r"\s+", " ", mm.group(1)).strip().replace(" *", "*")))
    j = re.search(r"struct DemczConfig\n(.*?)\nend", JL, flags=re.S).group(1)
    jfields = [tuple(x.strip().split("::")) for x in re.split(r"[;\n]", j) if x.strip()]
    want = {"int64_t": "Int64", "int32_t": "Int32", "uint64_t": "UInt64", "double": "Float64", "const int32_t*": "Ptr{Int32}",
            "const double*": "Ptr{Float64}", "void*": "Ptr{Cvoid}"}
    assert [f[0] for f in jfields] == [f[0] for f in cfields]
    for (jn, jt), (cn, ct) in zip(jfields, cfields):
        assert want[ct] == jt, f"{cn}: {ct} in the header, {jt} in the shim"


def test_closure_methods_exist_and_convert_block_index():
    """(f)1 / (f)4: the reference's defining feature -- an arbitrary `logobj` function (demcz.jl:189, README.md:14) --
    has methods in the shim, the 1-based block loop variable is converted for the 0-based ABI, and the 1-based
    `blockindex` entries are converted in `create`."""
    assert re.search(r"const LogObj = Union\{DeviceTarget,\s*Function\}", JL)
    assert re.search(r"function demcz_sample\(t::LogObj, Zmat, N=4", JL) and re.search(r"function demcz_anneal\(t::LogObj, Zmat, N=4", JL)
    assert re.search(r"demcz_sample\(t::LogObj, Zmat, opts::DEMCopt", JL) and re.search(r"demcz_anneal\(t::LogObj, Zmat, opts::DEMCopt", JL)
    assert "h, g, ib - 1, γ, Xprop" in JL                                 # demcz_propose takes the 0-based block
    assert "Int32[i - 1 for b in blockindex for i in b]" in JL            # blockindex entries 1-based -> 0-based
    assert re.search(r"lp\[ic\] = logobj\(Xp\[ic, :\]\)", JL)             # one closure call per chain and block-step
    assert "Int32(3)" in JL                                               # DEMCZ_TARGET_HOST_CALLBACK
    # demcopt's defaults are the reference's (DEMC.jl:41)
    assert "N=4, K=10, Ngeneration=5000, Nblocks=1, blockindex=[1:Npar], eps_scale=1e-4 * ones(Npar), γ=2.38" in JL
    assert "T0=3, TN=1e-3, autostop=:Rhat, autostop_every=1000, autostop_Rhat=1.05" in JL


def test_sharded_surface_diagnostics_and_checkpoint_are_bound():
    """VERDICT r2 'missing' #3: the multi-GPU entry points (the role of demcz_sample_par, src/demcz.jl:101-165), the reference's
    diagnostics by their own names (src/utils.jl:2-111) over the device reductions, and a checkpoint."""
    bound = {c[0] for c in julia_ccalls()}
    need = {"demcz_comm_unique_id", "demcz_comm_init", "demcz_set_append_lag", "demcz_set_comm_timeout", "demcz_synchronize",
            "demcz_rhat_array", "demcz_accept_ratio_array", "demcz_mean_cov_array", "demcz_mean_cov"}
    assert need <= bound, sorted(need - bound)
    assert re.search(r"function demcz_sample_par\(t::DeviceTarget, Zmat, opts::DEMCopt; sync_every=1000, prevrun=nothing, rank::Integer, nranks::Integer", JL)
    assert "chain_id0=c0" in JL and "comm_init(h, unique_id, nranks, rank)" in JL
    assert "Zmat[end-N+1+c0:end-N+c0+nloc, :]" in JL                       # rank r starts at ITS part of the last N rows (demcz.jl:113)
    for name in ("Rhat_gelman", "flatten_chain", "mean_cov_chain", "convergence_check", "save_checkpoint", "load_checkpoint"):
        assert re.search(rf"function {name}\(", JL), name
    # flatten_chain's column order: generation-major, then chain (utils.jl:26-29)
    assert "permutedims(chain[:, :, 1:Ngeneration], (2, 1, 3))" in JL


def test_flatten_chain_index_order_matches_the_reference_loop():
    """The permutedims/reshape form in the shim against the reference's explicit loop (utils.jl:22-32), in NumPy."""
    import numpy as np
    Npop, Npar, G = 3, 2, 4
    chain = np.arange(Npop * Npar * G, dtype=float).reshape((Npop, Npar, G), order="F")
    ref = np.zeros((Npar, Npop * G))
    for i in range(Npar):
        count = 0
        for ig in range(G):
            for ic in range(Npop):
                ref[i, count] = chain[ic, i, ig]
                count += 1
    # Julia: reshape(permutedims(chain, (2, 1, 3)), Npar, Npop*G) with column-major reshape
    jl = np.reshape(np.transpose(chain, (1, 0, 2)), (Npar, Npop * G), order="F")
    assert np.array_equal(jl, ref)
    import demc_jl_amd as demc
    assert np.array_equal(demc.flatten_chain(chain), ref)
