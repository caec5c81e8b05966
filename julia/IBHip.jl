# IBHip.jl -- the binding a maintainer of ImmersedBoundary.jl would add to run the per-partition
# residual hot path on MI355X through libibhip.so (C ABI: include/ibhip.h).
#
# It plugs into the reference's own hooks and nothing else:
#   * `conv_to_backend = IBHip.converter(dom)` (or `IBHip.hip`), `conv_from_backend = Array` in `dom(f, args...)`
#     and `impose_bc!` (/root/reference/src/ImmersedBoundary.jl:823-824, :846-855, :1202-1203);
#   * `ArrayBackends.to_backend(part, hip)` (src/arraybends.jl:14-77, registered for Partition at
#     src/ImmersedBoundary.jl:788) -- specialised here to upload ONCE and cache a native handle;
#   * Julia multiple dispatch on the operators (src/ImmersedBoundary.jl:873-1157): methods for
#     `HipPartition` + `HipArray` that `ccall` the library, including `divergent`, the tuple `face_gradient`, the
#     `Accumulator` call and `impose_bc!` on device-resident arrays;
#   * `Base.Broadcast` on `HipArray`: every broadcast node (`+ - * / max min abs`, scalars, `.=`, `.-=`, `@.`)
#     becomes one elementwise kernel (`ibh_ew_*`), so closures like test/advection.jl:67-83 run unchanged --
#     tests/test_gpu_broadcast.py runs exactly that expression tree through the same C entry points from Python.
#
# No KernelAbstractions, no CUDA.jl/AMDGPU.jl: device memory is owned by the library
# (ibh_malloc/ibh_free) behind `HipArray`.  There is no Julia runtime in the build container, so this
# file is written against the reference's sources and the C header but has never been executed; the same
# ABI is exercised from Python by tests/ (see INTEGRATION.md).
module IBHip

using ImmersedBoundary
import ImmersedBoundary: Partition, Boundary, Domain, Accumulator, at_owners, at_neighbors, at_faces, green_gauss,
    unsigned_green_gauss, cell_gradient, face_distance, owner_distance, neighbor_distance, face_gradient, MUSCL,
    divergent, impose_bc!
import ImmersedBoundary.CFD: JST_sensor
import ImmersedBoundary.ArrayBackends: to_backend
import ImmersedBoundary: CFD, Turbulence, Solver

const lib = get(ENV, "IBHIP_LIB", "libibhip")

@inline function check(rc::Cint)
    rc == 0 || error("libibhip: " * unsafe_string(ccall((:ibh_last_error, lib), Cstring, ())))
    nothing
end

init(device::Integer = 0) = check(ccall((:ibh_init, lib), Cint, (Cint,), device))

# ---------------------------------------------------------------------------------------------------
# device arrays (column-major like Julia's own arrays, so `(ncells, nvars)` maps 1:1 to the ABI's `ld`)
# ---------------------------------------------------------------------------------------------------
mutable struct HipArray{T, N} <: AbstractArray{T, N}
    ptr::Ptr{Cvoid}
    dims::NTuple{N, Int}
    parent::Any   # the array a column view aliases (kept alive), or nothing for an owning array
    function HipArray{T, N}(::UndefInitializer, dims::NTuple{N, Int}) where {T, N}
        p = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:ibh_malloc, lib), Cint, (Ptr{Ptr{Cvoid}}, Csize_t), p, max(prod(dims), 1) * sizeof(T)))
        a = new{T, N}(p[], dims, nothing)
        finalizer(x -> ccall((:ibh_free, lib), Cint, (Ptr{Cvoid},), x.ptr), a)
        a
    end
    # non-owning alias (no finalizer): `@view C[:, dim]`
    HipArray{T, N}(ptr::Ptr{Cvoid}, dims::NTuple{N, Int}, parent) where {T, N} = new{T, N}(ptr, dims, parent)
end
HipArray{T}(u::UndefInitializer, dims::Int...) where {T} = HipArray{T, length(dims)}(u, dims)
Base.size(a::HipArray) = a.dims
Base.similar(a::HipArray{T}, ::Type{S}, dims::Dims) where {T, S} = HipArray{S, length(dims)}(undef, dims)
Base.getindex(::HipArray, i...) = error("scalar indexing of a HipArray; copy it back with Array(a)")
# columns of a column-major (n, nv) field are contiguous: `@view C[:, dim]` / `C[:, dim]` alias them
Base.view(a::HipArray{T, 2}, ::Colon, j::Integer) where {T} =
    HipArray{T, 1}(a.ptr + (j - 1) * size(a, 1) * sizeof(T), (size(a, 1),), a)
Base.getindex(a::HipArray{T, 2}, ::Colon, j::Integer) where {T} = copy(view(a, :, j))

"`conv_to_backend`: host array -> device array."
function hip(a::Array{T, N}) where {T, N}
    d = HipArray{T, N}(undef, size(a))
    check(ccall((:ibh_h2d, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Csize_t), d.ptr, a, sizeof(a)))
    d
end
hip(a::AbstractArray) = hip(Array(a))
hip(a::HipArray) = a

"`conv_from_backend`: device array -> host array."
function Base.Array(d::HipArray{T, N}) where {T, N}
    a = Array{T, N}(undef, d.dims)
    check(ccall((:ibh_d2h, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Csize_t), a, d.ptr, sizeof(a)))
    a
end

ld(a::HipArray) = Int64(size(a, 1))
nv(a::HipArray) = Cint(length(a) ÷ max(size(a, 1), 1))

"The converter handed to `conv_to_backend`: callable like `hip`, and carries what `to_backend(::Partition, conv)` cannot
see in a `Partition` -- the mesh's block size (`dom.mesh.block_size`, src/mesher.jl:977), which switches the library's
block-structured fast path on (it still verifies every block against the face lists)."
struct HipConv
    block_size::Int
end
(c::HipConv)(a) = hip(a)
converter(dom::Domain) = HipConv(dom.mesh.block_size)

# ---------------------------------------------------------------------------------------------------
# Base.Broadcast: one elementwise kernel per node of the broadcast tree (include/ibhip.h, ibh_ew_*)
# ---------------------------------------------------------------------------------------------------
const EW_ADD, EW_SUB, EW_MUL, EW_DIV, EW_MAX, EW_MIN, EW_SUM = Cint.(0:6)
const EW_ABS, EW_NEG, EW_SQRT, EW_COPY = Cint.(16:19)
const _binop = IdDict{Any, Cint}(+ => EW_ADD, - => EW_SUB, * => EW_MUL, / => EW_DIV, max => EW_MAX, min => EW_MIN)
const _unop = IdDict{Any, Cint}(abs => EW_ABS, - => EW_NEG, sqrt => EW_SQRT, identity => EW_COPY, + => EW_COPY)

struct HipStyle <: Base.Broadcast.BroadcastStyle end
Base.Broadcast.BroadcastStyle(::Type{<:HipArray}) = HipStyle()
Base.Broadcast.BroadcastStyle(::HipStyle, ::Base.Broadcast.DefaultArrayStyle{0}) = HipStyle()   # scalars
Base.Broadcast.BroadcastStyle(::HipStyle, ::Base.Broadcast.BroadcastStyle) =
    error("broadcast between a HipArray and a host array: convert it with IBHip.hip first")

_rows(a::HipArray) = size(a, 1)
_operand(a::HipArray) = (a.ptr, nv(a), 0f0)
_operand(x::Number) = (C_NULL, Cint(0), Float32(x))
_operand(x::Base.RefValue) = _operand(x[])

"Evaluate one node: operands are HipArrays (same shape, or a column vector over the columns) or scalars."
function _ew(op, a, b, out::Union{HipArray, Nothing} = nothing)
    code = get(_binop, op) do
        error("IBHip broadcast: unsupported binary operation $op")
    end
    arrs = filter(x -> x isa HipArray, (a, b))
    isempty(arrs) && return op(a, b)
    big = arrs[argmax(map(length, arrs))]
    o = isnothing(out) ? similar(big) : out
    (pa, na, sa), (pb, nb, sb) = _operand(a), _operand(b)
    check(ccall((:ibh_ew_binary, lib), Cint,
        (Cint, Int64, Cint, Ptr{Cvoid}, Cint, Cfloat, Ptr{Cvoid}, Cint, Cfloat, Ptr{Cvoid}),
        code, _rows(big), nv(big), pa, na, sa, pb, nb, sb, o.ptr))
    o
end
function _ew(op, a::HipArray, out::Union{HipArray, Nothing} = nothing)
    code = get(_unop, op) do
        error("IBHip broadcast: unsupported unary operation $op")
    end
    o = isnothing(out) ? similar(a) : out
    check(ccall((:ibh_ew_unary, lib), Cint, (Cint, Int64, Ptr{Cvoid}, Ptr{Cvoid}), code, length(a), a.ptr, o.ptr))
    o
end

# recursive evaluation of a (possibly nested, `@.`-fused) Broadcasted tree; n-ary + * max min fold left like Julia does
# (`max.(a, b, 1f-7)` of the reference's MUSCL, src/ImmersedBoundary.jl:1113-1157)
_eval(x) = x
_eval(bc::Base.Broadcast.Broadcasted) = _node(bc.f, map(_eval, bc.args)...)
_node(f, a) = a isa HipArray ? _ew(f, a) : f(a)
_node(f, a, b) = _ew(f, a, b)
_node(f::Union{typeof(+), typeof(*), typeof(max), typeof(min)}, a, b, c, rest...) = _node(f, _node(f, a, b), c, rest...)

# ---- the whole (fused) Broadcasted tree as ONE launch: a postfix program for ibh_ew_eval (at most 48 instructions,
# 8 arrays, 8 scalars, stack depth 8); n-ary + and * fold left like Julia does.  `nothing` = does not fit / unsupported
# node: the caller falls back to the node-by-node evaluation above (same bits).
const EW_PUSH_ARRAY, EW_PUSH_SCALAR = Cint(32), Cint(33)
mutable struct _Prog
    code::Vector{Int32}
    arrs::Vector{HipArray}
    scal::Vector{Float32}
end
function _push!(P::_Prog, a::HipArray)
    k = findfirst(x -> x.ptr == a.ptr && size(x) == size(a), P.arrs)
    if isnothing(k)
        length(P.arrs) == 8 && return nothing
        push!(P.arrs, a)
        k = length(P.arrs)
    end
    push!(P.code, EW_PUSH_ARRAY | Int32((k - 1) << 8))
    1
end
function _push!(P::_Prog, x::Number)
    v = Float32(x)
    k = findfirst(==(v), P.scal)
    if isnothing(k)
        length(P.scal) == 8 && return nothing
        push!(P.scal, v)
        k = length(P.scal)
    end
    push!(P.code, EW_PUSH_SCALAR | Int32((k - 1) << 8))
    1
end
_push!(P::_Prog, x::Base.RefValue) = _push!(P, x[])
_push!(P::_Prog, x) = nothing
function _push!(P::_Prog, bc::Base.Broadcast.Broadcasted)
    f, args = bc.f, bc.args
    if length(args) == 1
        haskey(_unop, f) || return nothing
        d = _push!(P, args[1])
        isnothing(d) && return nothing
        push!(P.code, _unop[f])
        return d
    end
    (haskey(_binop, f) && (length(args) == 2 || f === (+) || f === (*) || f === max || f === min)) || return nothing
    depth = _push!(P, args[1])
    isnothing(depth) && return nothing
    for a in args[2:end]
        d = _push!(P, a)
        isnothing(d) && return nothing
        depth = max(depth, 1 + d)
        push!(P.code, _binop[f])
    end
    depth
end
"Evaluate `bc` into `dest` (or a new array) with one `ibh_ew_eval`; `nothing` if the tree does not fit one program."
function _fused(bc::Base.Broadcast.Broadcasted, dest::Union{HipArray, Nothing})
    P = _Prog(Int32[], HipArray[], Float32[])
    depth = _push!(P, bc)
    (isnothing(depth) || depth > 8 || length(P.code) > 48 || isempty(P.arrs)) && return nothing
    big = P.arrs[argmax(map(length, P.arrs))]
    n, k = _rows(big), nv(big)
    all(a -> _rows(a) == n && (nv(a) == k || nv(a) == 1), P.arrs) || return nothing
    o = isnothing(dest) ? similar(big) : dest
    (size(o, 1) == n && nv(o) == k) || return nothing
    ptrs = Ptr{Cvoid}[a.ptr for a in P.arrs]
    nvs = Int32[nv(a) for a in P.arrs]
    GC.@preserve P ptrs nvs check(ccall((:ibh_ew_eval, lib), Cint,
        (Int64, Cint, Cint, Ptr{Int32}, Cint, Ptr{Ptr{Cvoid}}, Ptr{Int32}, Cint, Ptr{Cfloat}, Ptr{Cvoid}),
        n, k, length(P.code), P.code, length(ptrs), ptrs, nvs, length(P.scal), P.scal, o.ptr))
    o
end

function Base.copy(bc::Base.Broadcast.Broadcasted{HipStyle})
    r = _fused(bc, nothing)
    isnothing(r) ? _eval(bc) : r
end
function Base.copyto!(dest::HipArray, bc::Base.Broadcast.Broadcasted{HipStyle})
    # `dest .= f.(dest, x)` / `dest .-= x`: one launch straight into dest (elementwise: dest may be an operand)
    isnothing(_fused(bc, dest)) || return dest
    if length(bc.args) == 2 && haskey(_binop, bc.f)
        a, b = map(_eval, bc.args)
        _ew(bc.f, a, b, dest)
    else
        r = _eval(bc)
        r isa HipArray ? _ew(identity, r, dest) : fill!(dest, r)
    end
    dest
end
Base.copyto!(dest::HipArray, bc::Base.Broadcast.Broadcasted{<:Base.Broadcast.DefaultArrayStyle{0}}) = fill!(dest, bc[])
function Base.fill!(a::HipArray{Float32}, v)
    check(ccall((:ibh_ew_fill, lib), Cint, (Int64, Cfloat, Ptr{Cvoid}), length(a), Float32(v), a.ptr))
    a
end
Base.copy(a::HipArray) = _ew(identity, a)

function _reduce(code::Cint, a::HipArray{Float32})
    out = HipArray{Float32, 1}(undef, (1,))
    check(ccall((:ibh_ew_reduce, lib), Cint, (Cint, Int64, Ptr{Cvoid}, Ptr{Cvoid}), code, length(a), a.ptr, out.ptr))
    Array(out)[1]
end
Base.maximum(a::HipArray{Float32}) = _reduce(EW_MAX, a)
Base.minimum(a::HipArray{Float32}) = _reduce(EW_MIN, a)
Base.sum(a::HipArray{Float32}) = _reduce(EW_SUM, a)

# ---------------------------------------------------------------------------------------------------
# Partition on the device: to_backend(part, hip) uploads once (the reference re-uploads per call, :848)
# ---------------------------------------------------------------------------------------------------
mutable struct HipPartition{Ti, Tf}   # (mutable: the handle is released by a finalizer, SURVEY.md 8b)
    handle::Ptr{Cvoid}
    host::Partition{Ti, Tf}
    nc::Int
    nf::Vector{Int}
    spacing::HipArray{Tf, 2}
    centers::HipArray{Tf, 2}
end
Base.ndims(p::HipPartition) = size(p.centers, 2)

const _cache = IdDict{Any, Any}()

"Bucketed `Accumulator.stencils` (src/accumulator.jl:46-61) -> CSR (offsets, indices), 1-based."
function csr(acc, n::Int)
    lens = zeros(Int32, n)
    for (len, (rows, _, _)) in acc.stencils
        lens[rows] .= len
    end
    off = Int32[1; 1 .+ cumsum(lens)]
    idx = Vector{Int32}(undef, off[end] - 1)
    for (len, (rows, st, _)) in acc.stencils
        for (k, r) in enumerate(rows), j = 1:len
            idx[off[r] + j - 1] = st[j, k]
        end
    end
    off, idx
end

to_backend(part::Partition, ::typeof(hip)) = to_backend(part, HipConv(8))   # mesher.jl:977 default; verified by the library
function to_backend(part::Partition{Ti, Tf}, conv::HipConv) where {Ti, Tf}
    get!(_cache, part) do
        nd = ndims(part)
        nc = size(part.spacing, 1)
        owners = [Int32.(part.face_owners_neighbors[d][1]) for d = 1:nd]
        neighs = [Int32.(part.face_owners_neighbors[d][2]) for d = 1:nd]
        left = [csr(part.face_accumulators[(d, false)], nc) for d = 1:nd]
        right = [csr(part.face_accumulators[(d, true)], nc) for d = 1:nd]
        nf = Int32[length(o) for o in owners]
        ptrs(v) = Ptr{Int32}[pointer(x) for x in v]
        h = Ref{Ptr{Cvoid}}(C_NULL)
        loff, lidx = first.(left), last.(left)
        roff, ridx = first.(right), last.(right)
        domain = Int32.(part.domain)
        iid = Int32.(part.image_in_domain)
        GC.@preserve owners neighs loff lidx roff ridx domain iid begin
            check(ccall((:ibh_partition_create, lib), Cint,
                (Ptr{Ptr{Cvoid}}, Cint, Int32, Ptr{Float32}, Ptr{Float32}, Ptr{Int32},
                 Ptr{Ptr{Int32}}, Ptr{Ptr{Int32}}, Ptr{Ptr{Int32}}, Ptr{Ptr{Int32}}, Ptr{Ptr{Int32}}, Ptr{Ptr{Int32}},
                 Int32, Ptr{Int32}, Ptr{Int32}, Cint, Cint),
                h, nd, nc, Array(part.spacing), Array(part.centers), nf,
                ptrs(owners), ptrs(neighs), ptrs(loff), ptrs(lidx), ptrs(roff), ptrs(ridx),
                length(iid), iid, domain, conv.block_size, 1 #= Julia indices =#))
        end
        hp = HipPartition{Ti, Tf}(h[], part, nc, Int.(nf), hip(Array(part.spacing)), hip(Array(part.centers)))
        finalizer(x -> ccall((:ibh_partition_destroy, lib), Cint, (Ptr{Cvoid},), x.handle), hp)
        hp
    end
end

# ---------------------------------------------------------------------------------------------------
# operators: same names / arity / return shapes as src/ImmersedBoundary.jl:873-1157
# ---------------------------------------------------------------------------------------------------
out_like(u::HipArray{T, N}, n::Int) where {T, N} = HipArray{T, N}(undef, (n, size(u)[2:end]...))

for (jl, c) in ((:at_owners, :ibh_at_owners), (:at_neighbors, :ibh_at_neighbors), (:at_faces, :ibh_at_faces),
                (:face_gradient, :ibh_face_gradient))
    @eval function $jl(part::HipPartition, u::HipArray, dim::Int)
        out = out_like(u, part.nf[dim])
        check(ccall(($(QuoteNode(c)), lib), Cint,
            (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Cint, Int64, Ptr{Cvoid}, Int64),
            part.handle, dim, u.ptr, nv(u), ld(u), out.ptr, ld(out)))
        out
    end
end

function cell_gradient(part::HipPartition, u::HipArray, dim::Int)
    out = out_like(u, part.nc)
    check(ccall((:ibh_cell_gradient, lib), Cint, (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Cint, Int64, Ptr{Cvoid}, Int64),
        part.handle, dim, u.ptr, nv(u), ld(u), out.ptr, ld(out)))
    out
end
# the tuple form (ImmersedBoundary.jl:980-988): all dimensions in one sweep per field (ibh_cell_gradient_nd); the
# gradients along dimension d are the columns (d-1)*nv+1 : d*nv of one (nc, nd*nv) buffer, returned as aliasing views
function cell_gradient(part::HipPartition, u::HipArray{Float32, N}) where {N}
    nd, n, k = ndims(part), part.nc, nv(u)
    buf = HipArray{Float32, 2}(undef, (n, nd * k))
    check(ccall((:ibh_cell_gradient_nd, lib), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Int64, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Int64),
        part.handle, u.ptr, k, ld(u), buf.ptr, n, C_NULL, 0))
    dims = N == 1 ? (n,) : (n, k)
    tuple((HipArray{Float32, N}(buf.ptr + (d - 1) * k * n * sizeof(Float32), dims, buf) for d = 1:nd)...)
end

function _gg(part::HipPartition, uf::HipArray, dim::Int, uns::Int)
    out = out_like(uf, part.nc)
    check(ccall((:ibh_green_gauss, lib), Cint, (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Cint, Int64, Ptr{Cvoid}, Int64, Cint),
        part.handle, dim, uf.ptr, nv(uf), ld(uf), out.ptr, ld(out), uns))
    out
end
green_gauss(part::HipPartition, uf::HipArray, dim::Int) = _gg(part, uf, dim, 0)
unsigned_green_gauss(part::HipPartition, uf::HipArray, dim::Int) = _gg(part, uf, dim, 1)

for (jl, c) in ((:face_distance, :ibh_face_distance), (:owner_distance, :ibh_owner_distance),
                (:neighbor_distance, :ibh_neighbor_distance))
    @eval function $jl(part::HipPartition{Ti, Tf}, dim::Int) where {Ti, Tf}
        out = HipArray{Tf, 1}(undef, (part.nf[dim],))
        check(ccall(($(QuoteNode(c)), lib), Cint, (Ptr{Cvoid}, Cint, Ptr{Cvoid}), part.handle, dim, out.ptr))
        out
    end
end

function JST_sensor(part::HipPartition, p::HipArray, dim::Int = 0)
    out = out_like(p, part.nc)
    check(ccall((:ibh_jst_sensor, lib), Cint, (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Cint, Int64, Ptr{Cvoid}, Int64),
        part.handle, dim, p.ptr, nv(p), ld(p), out.ptr, ld(out)))
    out
end

function MUSCL(part::HipPartition, u::HipArray, δu::HipArray, dim::Int;
               D::Union{HipArray, Nothing} = nothing, high_order::Bool = false)
    uL, uR = out_like(u, part.nf[dim]), out_like(u, part.nf[dim])
    check(ccall((:ibh_muscl, lib), Cint,
        (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Int64, Ptr{Cvoid}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Int64),
        part.handle, dim, u.ptr, δu.ptr, nv(u), ld(u), isnothing(D) ? C_NULL : D.ptr, high_order, uL.ptr, uR.ptr, ld(uL)))
    (uL, uR)
end

"`divergent(part, uf::Tuple)` (:950-956): sum over dims of `green_gauss`, accumulated in place."
function divergent(part::HipPartition, uf::Tuple)
    r = green_gauss(part, uf[1], 1)
    for dim = 2:ndims(part)
        r .+= green_gauss(part, uf[dim], dim)
    end
    r
end

"`face_gradient(part, u, ∇u::Tuple, dim)` (:1051-1069): normal component from the face difference, the others from `at_faces(∇u[i])`."
face_gradient(part::HipPartition, u::HipArray, ∇u::Tuple, dim::Int) =
    tuple((i == dim ? face_gradient(part, u, dim) : at_faces(part, ∇u[i], dim) for i = 1:ndims(part))...)

# ---------------------------------------------------------------------------------------------------
# Accumulator on the device (src/accumulator.jl:78-130, defaults op = +, f = identity, Δ = false: the only form the
# reference's callers use) -- interpolators of `impose_bc!`, coarseners / prolongators of `multigrid`
# ---------------------------------------------------------------------------------------------------
mutable struct HipAccumulator
    handle::Ptr{Cvoid}
    n_output::Int
    n_input::Int
end

"Bucketed stencils + weights -> CSR with weights (1-based indices)."
function csr_weighted(acc::Accumulator)
    off, idx = csr(acc, acc.n_output)
    w = ones(Float32, length(idx))
    for (len, (rows, _, wt)) in acc.stencils
        isnothing(wt) && continue
        for (k, r) in enumerate(rows), j = 1:len
            w[off[r] + j - 1] = wt[j, k]
        end
    end
    off, idx, w
end

function to_backend(acc::Accumulator, ::Union{typeof(hip), HipConv})
    get!(_cache, acc) do
        off, idx, w = csr_weighted(acc)
        n_in = isempty(idx) ? 0 : Int(maximum(idx))
        h = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:ibh_acc_create, lib), Cint,
            (Ptr{Ptr{Cvoid}}, Int32, Int32, Ptr{Int32}, Ptr{Int32}, Ptr{Float32}, Cint),
            h, acc.n_output, n_in, off, idx, w, 1))
        a = HipAccumulator(h[], acc.n_output, n_in)
        finalizer(x -> ccall((:ibh_acc_destroy, lib), Cint, (Ptr{Cvoid},), x.handle), a)
        a
    end
end

"`out .+= acc(a .- b)` in one launch (the prolongation step of `FAS!`, src/solver.jl:76)."
function accumulate_diff_add!(out::HipArray, acc::HipAccumulator, a::HipArray, b::HipArray)
    check(ccall((:ibh_accumulate_diff_add, lib), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Int64, Ptr{Cvoid}, Int64),
        acc.handle, a.ptr, b.ptr, nv(a), ld(a), out.ptr, ld(out)))
    out
end
function (acc::HipAccumulator)(v::HipArray)
    out = out_like(v, acc.n_output)
    check(ccall((:ibh_accumulate, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Int64, Ptr{Cvoid}, Int64),
        acc.handle, v.ptr, nv(v), ld(v), out.ptr, ld(out)))
    out
end

# ---------------------------------------------------------------------------------------------------
# impose_bc! on device-resident global arrays (src/ImmersedBoundary.jl:1197-1247): per boundary chunk
#   ia = W * a[image_domain]  (ibh_bc_interp),  ba = f(bdry, ia...),  a[ghost] = η ia + (1 - η) ba  (ibh_bc_blend)
# ---------------------------------------------------------------------------------------------------
mutable struct HipBoundary{Ti, Tf}
    handle::Ptr{Cvoid}
    host::Boundary{Ti, Tf}
    ng::Int
    projections::HipArray{Tf, 2}
    normals::HipArray{Tf, 2}
    image_distances::HipArray{Tf, 1}
    ghost_distances::HipArray{Tf, 1}
end

function to_backend(b::Boundary{Ti, Tf}, ::Union{typeof(hip), HipConv}) where {Ti, Tf}
    get!(_cache, b) do
        off, idx, w = csr_weighted(b.image_interpolator)
        gi, idm = Int32.(b.ghost_indices), Int32.(b.image_domain)
        h = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:ibh_bc_create, lib), Cint,
            (Ptr{Ptr{Cvoid}}, Int32, Ptr{Int32}, Ptr{Float32}, Ptr{Float32}, Int32, Ptr{Int32},
             Ptr{Int32}, Ptr{Int32}, Ptr{Float32}, Cint),
            h, length(gi), gi, Float32.(b.ghost_distances), Float32.(b.image_distances), length(idm), idm,
            off, idx, w, 1))
        hb = HipBoundary{Ti, Tf}(h[], b, length(gi), hip(Array(b.projections)), hip(Array(b.normals)),
                                 hip(Array(b.image_distances)), hip(Array(b.ghost_distances)))
        finalizer(x -> ccall((:ibh_bc_destroy, lib), Cint, (Ptr{Cvoid},), x.handle), hb)
        hb
    end
end

function impose_bc!(f, dom::Domain, bname::String, args::HipArray{Float32}...; kwargs...)
    for (_, b) in dom.boundaries[bname]
        bdry = to_backend(b, hip)
        iargs = map(args) do a
            ia = out_like(a, bdry.ng)
            check(ccall((:ibh_bc_interp, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Int64, Ptr{Cvoid}, Int64),
                bdry.handle, a.ptr, nv(a), ld(a), ia.ptr, ld(ia)))
            ia
        end
        r = f(bdry, iargs...; kwargs...)
        r isa Tuple || (r = (r,))
        for (a, ba, ia) in zip(args, r, iargs)
            if ba isa HipArray       # an array of boundary values, (ng,) broadcast over the columns or (ng, nv)
                bfull = size(ba) == size(ia) ? ba : ia .* 0f0 .+ ba
                check(ccall((:ibh_bc_blend, lib), Cint,
                    (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Int64, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Int64, Ptr{Cfloat}),
                    bdry.handle, a.ptr, nv(a), ld(a), ia.ptr, ld(ia), bfull.ptr, ld(bfull), C_NULL))
            else                     # a constant (per variable): the closure returned a number
                c = fill(Float32(ba), Int(nv(a)))
                check(ccall((:ibh_bc_blend, lib), Cint,
                    (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Int64, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Int64, Ptr{Cfloat}),
                    bdry.handle, a.ptr, nv(a), ld(a), ia.ptr, ld(ia), C_NULL, 0, c))
            end
        end
    end
    nothing
end

# ---------------------------------------------------------------------------------------------------
# fused residual sweeps (bypass broadcast: one C call = the whole closure of test/advection.jl:67-83)
# ---------------------------------------------------------------------------------------------------
"`ud .= -Σ_d green_gauss(upwind MUSCL/JST flux)`: the closure of test/advection.jl:67-83, fused."
function residual_advection!(ud::HipArray, part::HipPartition, u::HipArray, C::HipArray; flags::Integer = 0)
    check(ccall((:ibh_residual_advection, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Cint),
        part.handle, u.ptr, C.ptr, ld(C), ud.ptr, flags))
    ud
end

# layout of `ibh_fluid` (include/ibhip.h): R, gamma, mu_ref, Tref, S, nk, k[4]
struct IbhFluid
    R::Float32
    γ::Float32
    μref::Float32
    Tref::Float32
    S::Float32
    nk::Int32
    k::NTuple{4, Float32}
end
IbhFluid(fluid) = IbhFluid(fluid.R, fluid.γ, fluid.μref, fluid.Tref, fluid.S, Int32(min(length(fluid.k), 4)),
    ntuple(i -> i <= length(fluid.k) ? Float32(fluid.k[i]) : 0f0, 4))

"`R .= -Σ_d green_gauss(inviscid_fluxes(MUSCL(P)))` with the pressure JST sensor (CFD.inviscid_fluxes, cfd.jl:459)."
function residual_euler_hll!(R::HipArray, part::HipPartition, P::HipArray, fluid; flags::Integer = 0)
    f = Ref(IbhFluid(fluid))
    check(ccall((:ibh_residual_euler_hll, lib), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Int64, Ptr{IbhFluid}, Cint),
        part.handle, P.ptr, ld(P), R.ptr, ld(R), f, flags))
    R
end

# ---------------------------------------------------------------------------------------------------
# an explicit solver step, device resident: the body of `march!` (test/advection.jl:61-89)
# ---------------------------------------------------------------------------------------------------
"An ordered list of `impose_bc!` calls whose closures are `value` constants or `copy(u)` (test/advection.jl:30-46),
compiled once; `apply!(set, u)` has the semantics of the sequential calls."
mutable struct HipBCSet
    handle::Ptr{Cvoid}
    bcs::Vector{Any}          # keeps the HipBoundary structs alive
end
function HipBCSet(bcs::Vector, closures::Vector)   # closures[i] = :copy or a number
    modes = Int32[c === :copy ? 1 : 0 for c in closures]
    vals = Float32[c === :copy ? 0f0 : Float32(c) for c in closures]
    hs = Ptr{Cvoid}[b.handle for b in bcs]
    out = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve hs modes vals check(ccall((:ibh_bcset_create, lib), Cint,
        (Ptr{Ptr{Cvoid}}, Cint, Ptr{Ptr{Cvoid}}, Ptr{Int32}, Ptr{Cfloat}), out, length(hs), hs, modes, vals))
    s = HipBCSet(out[], collect(Any, bcs))
    finalizer(x -> ccall((:ibh_bcset_destroy, lib), Cint, (Ptr{Cvoid},), x.handle), s)
    s
end
apply!(s::HipBCSet, u::HipArray{Float32}) = (check(ccall((:ibh_bcset_apply, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}),
    s.handle, u.ptr)); u)

"`dt = scale * 0.5 / maximum(max.(unsigned_green_gauss(at_faces(C_d, d), d)...))` (test/advection.jl:52-59, 65) left in
device memory (`dt::HipArray` of length 1)."
function timestep_advection!(dt::HipArray{Float32}, part::HipPartition, C::HipArray{Float32}; scale = 1f0)
    check(ccall((:ibh_timestep_advection, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Cfloat, Ptr{Cvoid}),
        part.handle, C.ptr, ld(C), Float32(scale), dt.ptr))
    dt
end

"`u_out = u + dt * R(u)` (closure of test/advection.jl:67-83 + `u .+= ud .* dt`, :86) in one launch, then the BC set."
function step_advection!(u_out::HipArray{Float32}, part::HipPartition, u::HipArray{Float32}, C::HipArray{Float32},
                         dt::HipArray{Float32}, bcs::Union{HipBCSet, Nothing} = nothing)
    check(ccall((:ibh_step_advection, lib), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ptr{Cvoid}),
        part.handle, u.ptr, u_out.ptr, C.ptr, ld(C), dt.ptr, isnothing(bcs) ? C_NULL : bcs.handle))
    u_out
end

"The same step with `timestep_advection!(next_dt, part, C; scale)` for the NEXT step evaluated on the way (it depends on `C`
alone): extra workgroups of the BC set's own launches instead of two launches in front of the next sweep.  `next_dt` may be
`dt` itself."
function step_advection!(u_out::HipArray{Float32}, part::HipPartition, u::HipArray{Float32}, C::HipArray{Float32},
                         dt::HipArray{Float32}, bcs::Union{HipBCSet, Nothing}, next_dt::HipArray{Float32}; scale = 1f0)
    check(ccall((:ibh_step_advection_dt, lib), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Cfloat, Ptr{Cvoid}),
        part.handle, u.ptr, u_out.ptr, C.ptr, ld(C), dt.ptr, isnothing(bcs) ? C_NULL : bcs.handle, Float32(scale), next_dt.ptr))
    u_out
end

"`S .+ Σ_d green_gauss(at_faces(ν .+ νR, d) .* face_gradient(R, d) .- at_faces(vel[:, d] .* R, d), d)` in one launch
(the transport residual closed by `Wray_Agarwal`, src/turbulence.jl:222-241), bit-identical to the composition."
function scalar_transport!(out::HipArray{Float32}, part::HipPartition, R::HipArray{Float32}, νR::HipArray{Float32},
                           ν::Real, vel::HipArray{Float32}, S::HipArray{Float32})
    check(ccall((:ibh_scalar_transport, lib), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cfloat, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ptr{Cvoid}),
        part.handle, R.ptr, νR.ptr, Float32(ν), vel.ptr, ld(vel), S.ptr, out.ptr))
    out
end

"`shear_rate([cell_gradient(part, vel[:, i]) for i in 1:nd])` (src/turbulence.jl:110-124) in one launch where the partition
is made of complete 3-D blocks or has no block structure (an error otherwise: compose the operators)."
function shear_rate_of_velocity!(S::HipArray{Float32}, part::HipPartition, vel::HipArray{Float32})
    check(ccall((:ibh_shear_rate_of_velocity, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Cvoid}),
        part.handle, vel.ptr, ld(vel), S.ptr))
    S
end

"The same with the velocity gradients kept: `G` is `(nc, nd * nd)`, `∂u_i/∂x_j` in column `nd (j - 1) + i` -- the tuple
`cell_gradient(part, vel)` is the column blocks `G[:, nd (j - 1) + 1 : nd j]` -- for `viscous_residual!` (the gradients of a
Navier-Stokes closure with a turbulence model are made once)."
function shear_rate_of_velocity!(S::HipArray{Float32}, G::HipArray{Float32, 2}, part::HipPartition, vel::HipArray{Float32})
    check(ccall((:ibh_shear_rate_of_velocity_grad, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Int64),
        part.handle, vel.ptr, ld(vel), S.ptr, G.ptr, ld(G)))
    S, G
end

"`Wray_Agarwal(R, S, cell_gradient(part, R), cell_gradient(part, S))` (src/turbulence.jl:222-241) in one launch; returns
`(νt = nut, νR = nuR, S = Sout)` written into the three arrays."
function wray_agarwal_of!(nut::HipArray{Float32}, nuR::HipArray{Float32}, Sout::HipArray{Float32}, part::HipPartition,
                          R::HipArray{Float32}, S::HipArray{Float32}; σR = 0.72f0, C1 = 0.0829f0, κ = 0.41f0)
    check(ccall((:ibh_wray_agarwal_of, lib), Cint,
        (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cfloat, Cfloat, Cfloat, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
        part.handle, R.ptr, S.ptr, Float32(σR), Float32(C1), Float32(κ), nut.ptr, nuR.ptr, Sout.ptr))
    (νt = nut, νR = nuR, S = Sout)
end

# flags of the fused sweeps (include/ibhip.h)
const FORCE_GENERAL, IMAGE_ONLY, PASS_A_ONLY, PASS_B_ONLY, EXACT = 1, 2, 4, 8, 16
const PHASE_INTERIOR, PHASE_BOUNDARY, NO_FUSE, NO_QUAD = 32, 64, 128, 1024

# ---------------------------------------------------------------------------------------------------
# point-implicit smoother: the device kernels behind src/point_implicit.jl (hutchinson_trick :17-91,
# _inverse_blocks! :124-135, PIPreconditioner :141-161, proj_along / solve :221-329).  A maintainer adds
# methods of those functions for HipArray that call these; the Python mirror of this repo
# (immersedboundary.jl_amd/point_implicit.py) shows the composition one to one.
# ---------------------------------------------------------------------------------------------------
"`D .= pinv` of every `nv x nv` block of `D::(n, nv, nv)`, or `1 ./ (eps .+ D)` for a vector (:124-135)."
function inverse_blocks!(D::HipArray{Float32})
    n = size(D, 1); nv = ndims(D) == 1 ? 1 : size(D, 2)
    check(ccall((:ibh_pi_invert_blocks, lib), Cint, (Int64, Cint, Ptr{Cvoid}), n, nv, D.ptr))
    D
end

"`out[p, k] = Σ_i v[p, i] * invD[p, k, i]` (:153-161)."
function apply_blocks!(out::HipArray{Float32}, invD::HipArray{Float32}, v::HipArray{Float32})
    n = size(v, 1); nv = ndims(v) == 1 ? 1 : size(v, 2)
    check(ccall((:ibh_pi_apply_blocks, lib), Cint, (Int64, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
        n, nv, invD.ptr, v.ptr, out.ptr))
    out
end

"`s[p, k] += z[p] * (fxb[p, k] - fx[p, k]) / h`: one Hutchinson sample (:40)."
function hutch_accum!(s::HipArray{Float32}, fxb::HipArray{Float32}, fx::HipArray{Float32}, z::HipArray{Float32}, h)
    n = size(s, 1); nv = ndims(s) == 1 ? 1 : size(s, 2)
    check(ccall((:ibh_pi_hutch_accum, lib), Cint, (Int64, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cfloat, Ptr{Cvoid}),
        n, nv, fxb.ptr, fx.ptr, z.ptr, Float32(h), s.ptr))
    s
end

"`out = x + v*h` (:30, :112) and `out = (fxb - fx)/h` (:110-114)."
perturb!(out::HipArray{Float32}, x::HipArray{Float32}, v::HipArray{Float32}, h) = (check(ccall((:ibh_pi_perturb, lib), Cint,
    (Int64, Ptr{Cvoid}, Ptr{Cvoid}, Cfloat, Ptr{Cvoid}), length(x), x.ptr, v.ptr, Float32(h), out.ptr)); out)
fd!(out::HipArray{Float32}, fxb::HipArray{Float32}, fx::HipArray{Float32}, h) = (check(ccall((:ibh_pi_fd, lib), Cint,
    (Int64, Ptr{Cvoid}, Ptr{Cvoid}, Cfloat, Ptr{Cvoid}), length(fx), fxb.ptr, fx.ptr, Float32(h), out.ptr)); out)

"`x .+= s .* α; r .-= As .* α` with `α = dots[1] / (dots[2] + eps)` read on the device (:229-236, :291-294)."
function relax_update!(x::HipArray{Float32}, r::HipArray{Float32}, s::HipArray{Float32}, As::HipArray{Float32},
                       dots::Ptr{Cvoid})
    check(ccall((:ibh_pi_update, lib), Cint, (Int64, Ptr{Cvoid}, Cfloat, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
        length(r), dots, eps(Float32), s.ptr, As.ptr, x.ptr, r.ptr))
    nothing
end


# ---------------------------------------------------------------------------------------------------
# CFD pointwise physics on device arrays: methods of the reference's own functions (src/cfd.jl) for HipArray, so that a
# residual closure written against `CFD.*` runs unchanged after `conv_to_backend` (configs[1]-[4]).  Cartesian `dim` only
# (the matrix-normal forms of `inviscid_fluxes` / `viscous_fluxes` are a curvilinear extension outside this hot path).
# ---------------------------------------------------------------------------------------------------
_vec_like(a::HipArray{Float32}) = HipArray{Float32, ndims(a)}(undef, size(a))
for (jl, c) in ((:speed_of_sound, :ibh_cfd_speed_of_sound), (:dynamic_viscosity, :ibh_cfd_dynamic_viscosity),
                (:heat_conductivity, :ibh_cfd_heat_conductivity))
    @eval function CFD.$jl(fluid::CFD.Fluid, T::HipArray{Float32})   # cfd.jl:62-64, 71-77, 84-90
        out = _vec_like(T)
        f = Ref(IbhFluid(fluid))
        check(ccall(($(QuoteNode(c)), lib), Cint, (Ptr{IbhFluid}, Int64, Ptr{Cvoid}, Ptr{Cvoid}), f, length(T), T.ptr, out.ptr))
        out
    end
end
_ndP(P::HipArray{Float32, 2}) = size(P, 2) - 2
function CFD.primitive2state(fluid::CFD.Fluid, P::HipArray{Float32, 2})   # cfd.jl:106-123
    Q = _vec_like(P)
    f = Ref(IbhFluid(fluid))
    check(ccall((:ibh_cfd_primitive2state, lib), Cint, (Ptr{IbhFluid}, Cint, Int64, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Int64),
        f, _ndP(P), size(P, 1), P.ptr, ld(P), Q.ptr, ld(Q)))
    Q
end
function CFD.state2primitive(fluid::CFD.Fluid, Q::HipArray{Float32, 2})   # cfd.jl:137-151
    P = _vec_like(Q)
    f = Ref(IbhFluid(fluid))
    check(ccall((:ibh_cfd_state2primitive, lib), Cint, (Ptr{IbhFluid}, Cint, Int64, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Int64),
        f, _ndP(Q), size(Q, 1), Q.ptr, ld(Q), P.ptr, ld(P)))
    P
end
function CFD.inviscid_fluxes(fluid::CFD.Fluid, PL::HipArray{Float32, 2}, PR::HipArray{Float32, 2}, dim::Integer)   # :459-508
    F = _vec_like(PL)
    f = Ref(IbhFluid(fluid))
    check(ccall((:ibh_cfd_inviscid_fluxes_hll, lib), Cint,
        (Ptr{IbhFluid}, Cint, Cint, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Int64),
        f, _ndP(PL), dim, size(PL, 1), PL.ptr, PR.ptr, ld(PL), F.ptr, ld(F)))
    F
end
function CFD.inviscid_fluxes(fluid::CFD.Fluid, PL::HipArray{Float32, 2}, PR::HipArray{Float32, 2},
                             νL::HipArray{Float32, 1}, νR::HipArray{Float32, 1}, dim::Integer)                     # :516-554
    F = _vec_like(PL)
    f = Ref(IbhFluid(fluid))
    check(ccall((:ibh_cfd_inviscid_fluxes_sensor, lib), Cint,
        (Ptr{IbhFluid}, Cint, Cint, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64),
        f, _ndP(PL), dim, size(PL, 1), PL.ptr, PR.ptr, ld(PL), νL.ptr, νR.ptr, F.ptr, ld(F)))
    F
end
"`viscous_fluxes(fluid, P, Pgrad, dim; μₜ)` (cfd.jl:664-736): `Pgrad` a tuple / vector of the gradients of P along each axis."
function CFD.viscous_fluxes(fluid::CFD.Fluid, P::HipArray{Float32, 2}, Pgrad::Union{AbstractVector, Tuple}, dim::Integer;
                            μₜ::Union{HipArray{Float32, 1}, Real} = 0.0f0)
    F = _vec_like(P)
    f = Ref(IbhFluid(fluid))
    g = Ptr{Cvoid}[x.ptr for x in Pgrad]
    GC.@preserve g Pgrad begin
        check(ccall((:ibh_cfd_viscous_fluxes, lib), Cint,
            (Ptr{IbhFluid}, Cint, Cint, Int64, Ptr{Cvoid}, Int64, Ptr{Ptr{Cvoid}}, Int64, Ptr{Cvoid}, Cfloat, Ptr{Cvoid}, Int64),
            f, _ndP(P), dim, size(P, 1), P.ptr, ld(P), g, ld(Pgrad[1]),
            μₜ isa HipArray ? μₜ.ptr : C_NULL, μₜ isa HipArray ? 0f0 : Float32(μₜ), F.ptr, ld(F)))
    end
    F
end
"`R[:, 2:end] .+= Σ_d green_gauss(part, viscous_fluxes(fluid, at_faces(part, P, d), face_gradient(part, P, ∇P, d), d; μₜ = at_faces(part, μₜ, d)), d)`
in one launch, bit-identical to that composition (`∇P = cell_gradient(part, P)`)."
function viscous_residual!(R::HipArray{Float32, 2}, part::HipPartition, fluid::CFD.Fluid, P::HipArray{Float32, 2}, ∇P::Tuple,
                           μₜ::HipArray{Float32, 1})
    f = Ref(IbhFluid(fluid))
    g = Ptr{Cvoid}[x.ptr for x in ∇P]
    GC.@preserve g ∇P begin
        check(ccall((:ibh_viscous_residual, lib), Cint,
            (Ptr{Cvoid}, Ptr{IbhFluid}, Ptr{Cvoid}, Int64, Ptr{Ptr{Cvoid}}, Int64, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Int64),
            part.handle, f, P.ptr, ld(P), g, ld(∇P[1]), size(∇P[1], 2) == size(P, 2) ? 2 : 0, μₜ.ptr, R.ptr, ld(R)))
    end
    R
end
"`(bc::FlowBC)(P, normals; image_distances, du!dn, transpiration)` (cfd.jl:243-300) on image-point arrays of a `HipBoundary`."
function (bc::CFD.FlowBC)(P::HipArray{Float32, 2}, normals::HipArray{Float32, 2};
                          image_distances::Union{Nothing, HipArray{Float32, 1}} = nothing,
                          du!dn::Union{Nothing, HipArray{Float32, 1}} = nothing,
                          transpiration::Union{Real, HipArray{Float32, 1}} = 0.0f0)
    isnothing(du!dn) == isnothing(image_distances) ||
        throw(error("du!dn and image_distances must be passed together for BC imposition"))
    out = _vec_like(P)
    f = Ref(IbhFluid(bc.fluid))
    u∞ = Float32.(collect(bc.u∞))
    check(ccall((:ibh_cfd_flow_bc, lib), Cint,
        (Ptr{IbhFluid}, Cint, Int64, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Int64, Cfloat, Cfloat, Ptr{Float32}, Cint, Ptr{Cvoid},
         Ptr{Cvoid}, Cfloat, Ptr{Cvoid}, Ptr{Cvoid}, Int64),
        f, _ndP(P), size(P, 1), P.ptr, ld(P), normals.ptr, ld(normals), Float32(bc.p∞), Float32(bc.T∞), u∞,
        bc.normal_flow ? 1 : 0, isnothing(image_distances) ? C_NULL : image_distances.ptr,
        isnothing(du!dn) ? C_NULL : du!dn.ptr, transpiration isa Real ? Float32(transpiration) : 0f0,
        transpiration isa HipArray ? transpiration.ptr : C_NULL, out.ptr, ld(out)))
    out
end

# ---------------------------------------------------------------------------------------------------
# Turbulence closures (src/turbulence.jl) on device arrays.  Velocity gradients: the reference's Matrix of vectors,
# `velocity_gradient[i, j]` = ∂u_i/∂x_j -> an nd x nd table of device pointers, row-major g[(i-1) nd + j].
# ---------------------------------------------------------------------------------------------------
function _grad_table(g::AbstractMatrix)
    nd = size(g, 1)
    nd, Ptr{Cvoid}[g[i, j].ptr for i = 1:nd for j = 1:nd]
end
_wall_params(κ, C, A, β, βstar, D, A⁺, ω) = Float32[κ, C, A, β, βstar, D, A⁺, ω]
function Turbulence.wall_function(y::HipArray{Float32, 1}, u::HipArray{Float32, 1}, ν::HipArray{Float32, 1};
                                  κ::Real = 0.41f0, C::Real = 4.9f0, A::Real = 19.0f0, β::Real = 0.075f0,
                                  βstar::Real = 0.09f0, D::Real = 4.2f0, A⁺::Real = 360.0f0,
                                  ω_fixed_point::Real = 0.5f0, n_iter::Int = 20)                    # turbulence.jl:72-98
    o = ntuple(_ -> _vec_like(y), 6)
    par = _wall_params(κ, C, A, β, βstar, D, A⁺, ω_fixed_point)
    check(ccall((:ibh_turb_wall_function, lib), Cint,
        (Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float32}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid},
         Ptr{Cvoid}, Ptr{Cvoid}),
        length(y), y.ptr, u.ptr, ν.ptr, par, n_iter, o[1].ptr, o[2].ptr, o[3].ptr, o[4].ptr, o[5].ptr, o[6].ptr))
    (uτ = o[1], νₜ = o[2], k = o[3], ω = o[4], ϵ = o[5], du!dn = o[6])
end
function Turbulence.wall_function(Rey::HipArray{Float32, 1};
                                  κ::Real = 0.41f0, C::Real = 4.9f0, A::Real = 19.0f0, β::Real = 0.075f0,
                                  βstar::Real = 0.09f0, D::Real = 4.2f0, A⁺::Real = 360.0f0,
                                  ω_fixed_point::Real = 0.5f0, n_iter::Int = 20)                    # turbulence.jl:27-70
    o = ntuple(_ -> _vec_like(Rey), 5)
    par = _wall_params(κ, C, A, β, βstar, D, A⁺, ω_fixed_point)
    check(ccall((:ibh_turb_wall_function_rey, lib), Cint,
        (Int64, Ptr{Cvoid}, Ptr{Float32}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
        length(Rey), Rey.ptr, par, n_iter, o[1].ptr, o[2].ptr, o[3].ptr, o[4].ptr, o[5].ptr))
    (y⁺ = o[1], u⁺ = o[2], μ⁺ = o[3], k⁺ = o[4], du⁺!dy⁺ = o[5])
end
function Turbulence.shear_rate(velocity_gradient::AbstractMatrix{<:HipArray})                       # :110-124
    nd, tab = _grad_table(velocity_gradient)
    S = _vec_like(velocity_gradient[1, 1])
    GC.@preserve tab velocity_gradient check(ccall((:ibh_turb_shear_rate, lib), Cint,
        (Cint, Int64, Ptr{Ptr{Cvoid}}, Ptr{Cvoid}), nd, length(S), tab, S.ptr))
    S
end
function Turbulence.Smagorinsky_νSGS(Δ::HipArray{Float32, 1}, S::HipArray{Float32, 1}; Cₛ::Real = 0.17f0)   # :134-138
    out = _vec_like(S)
    check(ccall((:ibh_turb_smagorinsky, lib), Cint, (Int64, Ptr{Cvoid}, Ptr{Cvoid}, Cfloat, Ptr{Cvoid}),
        length(S), Δ.ptr, S.ptr, Float32(Cₛ), out.ptr))
    out
end
function Turbulence.standard_kϵ(k::HipArray{Float32, 1}, ϵ::HipArray{Float32, 1}, S::HipArray{Float32, 1};
                                Cμ::Real = 0.09f0, σk::Real = 1.0f0, σϵ::Real = 1.3f0, C1ϵ::Real = 1.44f0,
                                C2ϵ::Real = 1.92f0)                                                 # :175-196
    o = ntuple(_ -> _vec_like(k), 5)
    par = Float32[Cμ, σk, σϵ, C1ϵ, C2ϵ]
    check(ccall((:ibh_turb_k_epsilon, lib), Cint,
        (Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float32}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
        length(k), k.ptr, ϵ.ptr, S.ptr, par, o[1].ptr, o[2].ptr, o[3].ptr, o[4].ptr, o[5].ptr))
    (νk = o[1], νϵ = o[2], Sk = o[3], Sϵ = o[4], νₜ = o[5])
end
function Turbulence.Wray_Agarwal(R::HipArray{Float32, 1}, S::HipArray{Float32, 1}, ∇R::HipArray{Float32, 2},
                                 ∇S::HipArray{Float32, 2}; σR::Real = 0.72f0, C₁::Real = 0.0829f0, κ::Real = 0.41f0)   # :222-241
    o = ntuple(_ -> _vec_like(R), 3)
    check(ccall((:ibh_turb_wray_agarwal, lib), Cint,
        (Cint, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Cvoid}, Int64, Cfloat, Cfloat, Cfloat, Ptr{Cvoid},
         Ptr{Cvoid}, Ptr{Cvoid}),
        size(∇R, 2), length(R), R.ptr, S.ptr, ∇R.ptr, ld(∇R), ∇S.ptr, ld(∇S), Float32(σR), Float32(C₁), Float32(κ),
        o[1].ptr, o[2].ptr, o[3].ptr))
    (νₜ = o[1], νR = o[2], S = o[3])
end
function Turbulence.Ducros_sensor(velocity_gradient::AbstractMatrix{<:HipArray})                    # :253-282
    nd, tab = _grad_table(velocity_gradient)
    out = _vec_like(velocity_gradient[1, 1])
    GC.@preserve tab velocity_gradient check(ccall((:ibh_turb_ducros, lib), Cint,
        (Cint, Int64, Ptr{Ptr{Cvoid}}, Ptr{Cvoid}), nd, length(out), tab, out.ptr))
    out
end
function Turbulence.WALE_νSGS(Δ::HipArray{Float32, 1}, velocity_gradient::AbstractMatrix{<:HipArray}; Cw::Real = 0.325f0)   # :292-337
    nd, tab = _grad_table(velocity_gradient)
    @assert nd == 3 "WALE model only implemented for 3D"
    out = _vec_like(Δ)
    GC.@preserve tab velocity_gradient check(ccall((:ibh_turb_wale, lib), Cint,
        (Int64, Ptr{Cvoid}, Ptr{Ptr{Cvoid}}, Cfloat, Ptr{Cvoid}), length(Δ), Δ.ptr, tab, Float32(Cw), out.ptr))
    out
end

# ---------------------------------------------------------------------------------------------------
# Solver.FAS! (src/solver.jl:39-91) on device arrays: the reference's loop with its array passes as library calls --
# `r .+= source; Q .+= clamp(ω, 0, 1) .* r; norm(r)` is ONE launch (ibh_fas_update), the prolongation step another
# (accumulate_diff_add!).  Same keyword arguments, same quirks (recursion guard `length(coarseners) > 1`), same return value.
# ---------------------------------------------------------------------------------------------------
const _norm2 = Ref{Ptr{Cvoid}}(C_NULL)
function _dscalar()
    if _norm2[] == C_NULL
        p = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:ibh_malloc, lib), Cint, (Ptr{Ptr{Cvoid}}, Csize_t), p, 8))
        _norm2[] = p[]
    end
    _norm2[]
end
"`rr = r [.+ source]; [Q .+= clamp(ω, 0, 1) .* rr]`; returns `norm(rr)`."
function _fas_pass(r::HipArray{Float32}, source, Q, ω::Real)
    d = _dscalar()
    check(ccall((:ibh_fas_update, lib), Cint, (Int64, Cfloat, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
        length(r), Float32(ω), r.ptr, isnothing(source) ? C_NULL : source.ptr, isnothing(Q) ? C_NULL : Q.ptr, d))
    h = Ref{Float64}(0.0)
    check(ccall((:ibh_d2h, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Csize_t), h, d, 8))
    Float32(sqrt(h[]))
end
function Solver.FAS!(f, Q::HipArray{Float32};
                     coarseners = [], prolongators = [],
                     perscribed_f::Union{HipArray, Nothing} = nothing,
                     multigrid_level::Int = 0, n_iter::Int = 50, rtol::Real = 1.0f-1, atol::Real = 1.0f-7)
    l = multigrid_level
    fQ, ω = f(l, Q)
    source = isnothing(perscribed_f) ? nothing : perscribed_f .- fQ
    nr0 = _fas_pass(fQ, source, nothing, 0f0)
    nr = nr0
    if length(coarseners) > 1
        coars, prolong = to_backend(coarseners[1], hip), to_backend(prolongators[1], hip)
        Qc = coars(Q)
        Qcold = copy(Qc)
        pfQc = coars(isnothing(source) ? fQ : fQ .+ source)
        Solver.FAS!(f, Qc; coarseners = coarseners[2:end], prolongators = prolongators[2:end], perscribed_f = pfQc,
                    multigrid_level = l + 1, n_iter = n_iter, atol = atol, rtol = rtol)
        accumulate_diff_add!(Q, prolong, Qc, Qcold)                  # Q .+= prolongators[1](Qc .- Qcold)
    end
    for _ = 1:n_iter
        r, ω = f(l, Q)
        nr = _fas_pass(r, source, Q, ω)                              # r .+= source; Q .+= clamp(ω, 0, 1) .* r; norm(r)
        nr < nr0 * rtol + atol && break
    end
    nr / (nr0 + eps(Float32))
end
to_backend(a::HipAccumulator, ::Union{typeof(hip), HipConv}) = a

import LinearAlgebra
"`norm(a)` of a device array (what `FAS!` calls, src/solver.jl:57,84): sum of squares in Float64 on the device."
function LinearAlgebra.norm(a::HipArray{Float32})
    d = _dscalar()
    check(ccall((:ibh_sumsq, lib), Cint, (Int64, Ptr{Cvoid}, Ptr{Cvoid}), length(a), a.ptr, d))
    h = Ref{Float64}(0.0)
    check(ccall((:ibh_d2h, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Csize_t), h, d, 8))
    Float32(sqrt(h[]))
end

end # module
