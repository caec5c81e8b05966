"""julia/IBHip.jl cannot be executed here (no Julia runtime in the build container or on the GPU box); what CAN be checked
is that every ``ccall`` in it names a function include/ibhip.h declares, with the same number of arguments and the same
kind of argument in every position (pointer / integer of the same width / float / double) and the same return type --
the class of mistake a first run under Julia would otherwise find one crash at a time.  Also: every handle the binding
creates has its destroy call in a finalizer (SURVEY.md 8b "wrapper frees in finalizers")."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JL = os.path.join(ROOT, "julia", "IBHip.jl")
HDR = os.path.join(ROOT, "include", "ibhip.h")

C_KIND = {"int": "i32", "int32_t": "i32", "uint32_t": "u32", "unsigned": "u32", "int64_t": "i64", "uint64_t": "u64",
          "size_t": "u64", "float": "f32", "double": "f64"}
JL_KIND = {"Cint": "i32", "Int32": "i32", "UInt32": "u32", "Cuint": "u32", "Int64": "i64", "UInt64": "u64", "Csize_t": "u64",
           "Cfloat": "f32", "Float32": "f32", "Cdouble": "f64", "Float64": "f64", "Cstring": "ptr"}


def header_prototypes():
    s = open(HDR).read()
    s = re.sub(r"/\*.*?\*/", " ", s, flags=re.S)
    s = re.sub(r"//[^\n]*", " ", s)
    s = re.sub(r"^\s*#[^\n]*", " ", s, flags=re.M)
    protos = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(ibh_\w+)\s*\(([^;{}]*?)\)\s*;", s):
        ret, name, args = m.group(1).strip(), m.group(2), " ".join(m.group(3).split())
        if "typedef" in ret or "struct" in ret and "(" in ret:
            continue
        kinds = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a or "[" in a:
                    kinds.append("ptr")
                    continue
                toks = [t for t in re.split(r"\s+", a) if t not in ("const", "struct")]
                ty = toks[0]
                assert ty in C_KIND, f"{name}: unknown C type in `{a}`"
                kinds.append(C_KIND[ty])
        rk = "ptr" if "*" in ret else C_KIND.get(ret.replace("const", "").strip(), ret)
        protos[name] = (rk, kinds)
    return protos


def julia_ccalls():
    s = open(JL).read()
    s = re.sub(r"#=.*?=#", " ", s, flags=re.S)
    s = "\n".join(l.split("#")[0] if "ccall" not in l.split("#")[0] or True else l for l in s.split("\n"))
    calls = []
    pat = re.compile(r"ccall\(\((:\w+|\$\(QuoteNode\(c\)\)),\s*lib\),\s*(\w+),\s*\(([^()]*)\)", re.S)
    for m in pat.finditer(s):
        sym, ret, types = m.group(1), m.group(2), m.group(3)
        kinds = []
        for t in [x.strip() for x in types.split(",") if x.strip()]:
            kinds.append("ptr" if t.startswith("Ptr{") or t.startswith("Ref{") else JL_KIND.get(t, "?" + t))
        if sym.startswith("$"):   # generated methods: the `for (jl, c) in ((:at_owners, :ibh_at_owners), ...)` table above
            head = s[:m.start()]
            tab = head[head.rindex("for (jl, c) in"):]
            names = re.findall(r"\(:\w+,\s*:(ibh_\w+)\)", tab[:tab.index("@eval")])
            assert names
        else:
            names = [sym[1:]]
        for n in names:
            calls.append((n, JL_KIND.get(ret, "ptr" if ret.startswith("Ptr") else "?" + ret), kinds, s[:m.start()].count("\n") + 1))
    return calls


def test_every_ccall_matches_the_header():
    protos = header_prototypes()
    assert len(protos) >= 100
    calls = julia_ccalls()
    assert len(calls) >= 60
    bad = []
    for name, ret, kinds, line in calls:
        if name not in protos:
            bad.append(f"IBHip.jl:{line}: {name} is not declared in include/ibhip.h")
            continue
        cret, ckinds = protos[name]
        if ret != cret:
            bad.append(f"IBHip.jl:{line}: {name} returns {cret} in the header, {ret} in the ccall")
        if len(kinds) != len(ckinds):
            bad.append(f"IBHip.jl:{line}: {name} takes {len(ckinds)} arguments, the ccall passes {len(kinds)}")
            continue
        for i, (a, b) in enumerate(zip(kinds, ckinds)):
            if a != b:
                bad.append(f"IBHip.jl:{line}: {name} argument {i + 1}: header {b}, ccall {a}")
    assert not bad, "\n".join(bad)


def test_binding_covers_the_closures_of_every_config():
    """configs[1]-[4] need more than the grid operators: the CFD / turbulence kernels, FAS!'s update, the fused entries."""
    bound = {c[0] for c in julia_ccalls()}
    need = {"ibh_cfd_speed_of_sound", "ibh_cfd_dynamic_viscosity", "ibh_cfd_heat_conductivity", "ibh_cfd_primitive2state",
            "ibh_cfd_state2primitive", "ibh_cfd_inviscid_fluxes_hll", "ibh_cfd_inviscid_fluxes_sensor",
            "ibh_cfd_viscous_fluxes", "ibh_cfd_flow_bc", "ibh_turb_wall_function", "ibh_turb_shear_rate",
            "ibh_turb_smagorinsky", "ibh_turb_k_epsilon", "ibh_turb_wray_agarwal", "ibh_turb_ducros", "ibh_turb_wale",
            "ibh_fas_update", "ibh_sumsq", "ibh_viscous_residual", "ibh_partition_destroy", "ibh_acc_destroy",
            "ibh_bc_destroy", "ibh_residual_euler_hll", "ibh_residual_advection", "ibh_accumulate", "ibh_bc_interp",
            "ibh_bc_blend", "ibh_step_advection", "ibh_step_advection_dt", "ibh_timestep_advection",
            "ibh_shear_rate_of_velocity", "ibh_shear_rate_of_velocity_grad", "ibh_wray_agarwal_of", "ibh_scalar_transport"}
    assert not (need - bound), sorted(need - bound)


def test_handles_are_freed_in_finalizers():
    s = open(JL).read()
    for struct, destroy in (("HipPartition", "ibh_partition_destroy"), ("HipAccumulator", "ibh_acc_destroy"),
                            ("HipBoundary", "ibh_bc_destroy"), ("HipBCSet", "ibh_bcset_destroy")):
        assert re.search(r"mutable struct " + struct + r"\b", s), f"{struct} must be a mutable struct to carry a finalizer"
        assert re.search(r"finalizer\([^\n]*:" + destroy, s), f"no finalizer calls {destroy}"
