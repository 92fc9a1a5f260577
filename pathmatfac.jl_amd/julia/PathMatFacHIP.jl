# PathMatFacHIP.jl -- Julia host shim: re-points PathMatFac's `mf_fit!` (src/fit.jl:9-38), the only caller of the
# un-vendored MatFac.fit! inner loop, at libpmf_hip.so (include/pmf_hip.h) through `ccall`.
#
# Everything above `mf_fit!` in src/fit.jl (mf_fit_adapt_lr!, init_theta!, init_factors!, fit_ard!,
# fit_feature_set_ard!, fit!, transform) runs unchanged: it only sees the history Dict this function returns
# ("term_code", "epochs", src/fit.jl:63,69) and the mutated model parameters.
#
# Usage:   using PathMatFac; include("PathMatFacHIP.jl"); PathMatFacHIP.install!("/path/to/libpmf_hip.so")
#          model = PathMatFacModel(D; ...);  fit!(model; ...)         # no gpu(model): the library owns the device copy
#
# NOTE: Julia is not installed in the build container, so this file has not been executed there; it is the
# maintainer-side binding that INTEGRATION.md documents.  The Python ctypes binding (_lib.py) is its tested twin.
module PathMatFacHIP

import PathMatFac
const PM = PathMatFac

const LIB = Ref{String}("libpmf_hip.so")
const CTX = IdDict{Any,Ptr{Cvoid}}()          # model => pmf_ctx*
const DATA_KEY = IdDict{Any,UInt}()           # model => objectid(model.data) last uploaded
const OPT_KEY = IdDict{Any,UInt}()            # model => objectid(opt) whose state lives on the device

const TERM_CODES = ("max_epochs", "loss_increase", "abs_tol", "rel_tol", "nonfinite")
const NOISE_KIND = Dict("normal" => Cint(0), "bernoulli" => Cint(1), "poisson" => Cint(2))

struct FitOpts            # pmf_fit_opts
    update_X::Cint; update_Y::Cint; update_col_layers::Cint; frozen_layers::Cint; frozen_regs::Cint
    max_epochs::Cint; epoch::Cint; tol_max_iters::Cint; keep_trace::Cint; verbosity::Cint; print_iter::Cint
    reserved::Cint; abs_tol::Cdouble; rel_tol::Cdouble; capacity::Int64
end

mutable struct FitResult  # pmf_fit_result
    term_code::Cint; epochs::Cint; n_trace::Cint; trace_cap::Cint
    final_loss::Cdouble; loss_trace::Ptr{Cdouble}; seconds::Cdouble
end

lasterr() = unsafe_string(ccall((:pmf_last_error, LIB[]), Cstring, ()))
chk(rc::Integer) = rc == 0 ? nothing : error("libpmf_hip: " * lasterr())
f32(a) = convert(Array{Float32}, a)
starts(rs) = Int64[r.start for r in rs]
stops(rs) = Int64[r.stop for r in rs]

# Arithmetic of the data pass's matrix products (include/pmf_hip.h): :f32 = exact f32 MFMA (default),
# :bf16x3 = split-bf16 products where a kernel variant exists (DESIGN.md 4.5).  No reference counterpart.
function set_precision!(model, mode::Symbol)
    code = mode == :f32 ? 0 : mode == :bf16x3 ? 1 : error("precision must be :f32 or :bf16x3")
    chk(ccall((:pmf_set_precision, LIB[]), Cint, (Ptr{Cvoid}, Cint), context!(model), code))
end

# Storage type of the device copy of D (include/pmf_hip.h): :f32 (default) or :bf16 (BASELINE configs[4], "D stored
# bf16": rounded once at upload, everything else stays Float32).  Takes effect at the next upload of the data matrix.
const STORE = Ref{Cint}(0)
set_store!(mode::Symbol) = (STORE[] = mode == :f32 ? 0 : mode == :bf16 ? 1 : error("store must be :f32 or :bf16"); nothing)

# ---- multi-GPU: one Julia process per GPU (analyses/scripts/julia/script_util.jl:278-306), rows sharded ----------------
# Rank 0 creates the 128-byte RCCL id and hands it to the other ranks by any means (a file, MPI, Distributed.jl):
#     id = PathMatFacHIP.comm_unique_id()                       # rank 0
#     PathMatFacHIP.comm_init!(model, rank, nranks, id)         # every rank, before fit!
# From then on mf_fit! all-reduces grad(Y), the loss and the layer gradients inside pmf_fit (DESIGN.md 5); the
# statistics the host keeps between the GD stages go through comm_allreduce!.
# HIP devices visible to this process (pmf_device_count).  A launcher that pins one device per rank leaves ONE visible device,
# which is device 0 whatever the local rank: device_for_rank mirrors bench.py's map.
function device_count()
    n = Ref{Cint}(0)
    chk(ccall((:pmf_device_count, LIB[]), Cint, (Ref{Cint},), n))
    return Int(n[])
end
device_for_rank(local_rank::Integer) = (v = device_count(); v == 1 ? 0 : (local_rank < v ? Int(local_rank) :
    error("local rank $local_rank has no device: only $v visible")))
# kernel family of the last fused data pass (pmf_debug_last_kernel): 0 exact f32, 1 / 2 / 4 / 8 = split kernels sb / sb2 / sb4 / sb8
function last_kernel(model)
    k = Ref{Cint}(0)
    chk(ccall((:pmf_debug_last_kernel, LIB[]), Cint, (Ptr{Cvoid}, Ref{Cint}), context!(model), k))
    return Int(k[])
end

function comm_unique_id()
    id = zeros(UInt8, 128)
    GC.@preserve id chk(ccall((:pmf_comm_get_unique_id, LIB[]), Cint, (Ptr{UInt8},), id))
    return id
end
function comm_init!(model, rank::Integer, nranks::Integer, id::Vector{UInt8}; device::Integer=rank)
    ctx = context!(model; device=device)
    GC.@preserve id chk(ccall((:pmf_comm_init, LIB[]), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{UInt8}), ctx, rank, nranks, id))
end
comm_destroy!(model) = chk(ccall((:pmf_comm_destroy, LIB[]), Cint, (Ptr{Cvoid},), context!(model)))
comm_set_chunks!(model, n::Integer) = chk(ccall((:pmf_comm_set_chunks, LIB[]), Cint, (Ptr{Cvoid}, Cint), context!(model), n))
# sum (op = 0) or maximum (op = 1) over the ranks of a host array, in place (Float32 or Float64)
function comm_allreduce!(model, a::Array{T}; op::Integer=0) where {T<:Union{Float32,Float64}}
    GC.@preserve a chk(ccall((:pmf_comm_allreduce, LIB[]), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Cint, Cint),
                             context!(model), a, length(a), T == Float64 ? 1 : 0, op))
    return a
end

# Adopt the host's HIP stream (AMDGPU.jl: AMDGPU.stream().stream); C_NULL = the library's own non-blocking stream.
set_stream!(model, stream::Ptr{Cvoid}) = chk(ccall((:pmf_set_stream, LIB[]), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), context!(model), stream))

# update_A! of one FeatureSetARDReg view on the device (src/featureset_ard.jl:214-294): S is the view's L x N_v feature-set
# matrix (dense, row-major for the library = the transpose of Julia's column-major N_v x L), A / ssq_grad are K x L here
# (factor index contiguous).  Returns (best_loss, epochs_run); beta[:, cr] is updated on the device and returned in beta.
function fsard_update_A!(model, cr::UnitRange, S_t::Matrix{Float32}, alpha::Vector{Float32}, lambda::Vector{Float32},
                         A_t::Matrix{Float32}, ssq_t::Matrix{Float32}, beta::Matrix{Float32};
                         alpha0::Float32, v0::Float32, lr::Float32, max_epochs::Integer=1000, term_iter::Integer=50,
                         atol::Float64=1e-5)
    best = Ref{Cdouble}(0.0); ep = Ref{Cint}(0)
    L = size(S_t, 2)
    GC.@preserve S_t alpha lambda A_t ssq_t beta chk(ccall((:pmf_fsard_update_A, LIB[]), Cint,
        (Ptr{Cvoid}, Int64, Int64, Cint, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Cfloat, Cfloat, Cfloat, Ptr{Cfloat},
         Ptr{Cfloat}, Cint, Cint, Cdouble, Ref{Cdouble}, Ref{Cint}, Ptr{Cfloat}),
        context!(model), cr.start, cr.stop, L, S_t, alpha, lambda, alpha0, v0, lr, ssq_t, A_t, max_epochs, term_iter, atol,
        best, ep, beta))
    return best[], Int(ep[])
end

function context!(model; device::Integer=0)
    ctx = get!(CTX, model) do
        p = Ref{Ptr{Cvoid}}(C_NULL)
        chk(ccall((:pmf_create, LIB[]), Cint, (Cint, Ref{Ptr{Cvoid}}), device, p))
        p[]
    end
    key = objectid(model.data)
    if get(DATA_KEY, model, UInt(0)) != key            # gpu(model): upload the data matrix once
        D = f32(model.data)
        M, N = size(D)
        GC.@preserve D chk(ccall((:pmf_set_data, LIB[]), Cint, (Ptr{Cvoid}, Ptr{Cfloat}, Int64, Int64, Cint),
                                 ctx, D, M, N, STORE[]))
        DATA_KEY[model] = key
    end
    return ctx
end

# MatFac noise structs -> the ABI's names: NormalNoise -> "normal", BernoulliNoise -> "bernoulli", PoissonNoise -> "poisson".
# (MatFac.jl is un-vendored: the per-column weight field set by MF.set_weight! (src/fit.jl:157,180) is assumed to be
#  `weight`; adjust this one accessor if the struct names it differently.)
noise_name(n) = lowercase(replace(string(nameof(typeof(n))), "Noise" => ""))
noise_weights(nm) = vcat([collect(n.weight) for n in nm.noises]...)

unwrap(l) = isa(l, PM.FrozenLayer) ? l.layer : l
unwrapreg(r) = isa(r, PM.FrozenRegularizer) ? r.reg : r

# one-hot CSC row_batches matrix -> 0-based batch index of every row (src/util.jl:200-210, 588-593)
function batch_of_row(rb)
    M, nb = size(rb)
    out = fill(Int32(-1), M)
    for b in 1:nb, i in PM.get_col_idx(rb, b)
        out[i] = b - 1
    end
    return out
end

function add_reg!(ctx, which::Symbol, reg, p::Float32=1f0)
    sym(s) = Symbol("pmf_add_", which == :X ? "xreg_" : "yreg_", s)
    if isa(reg, Function)                               # x -> 0
        return
    elseif isa(reg, PM.L2Regularizer)
        w = f32(reg.weights)
        GC.@preserve w chk(ccall((sym("l2"), LIB[]), Cint, (Ptr{Cvoid}, Ptr{Cfloat}, Cfloat), ctx, w, p))
    elseif isa(reg, PM.GroupRegularizer)
        s, e = starts(reg.group_idx), stops(reg.group_idx)
        w = f32(hcat(reg.group_weights...))             # K x n_groups column-major == n_groups x K with K contiguous
        GC.@preserve s e w chk(ccall((sym("group"), LIB[]), Cint,
            (Ptr{Cvoid}, Cint, Ptr{Int64}, Ptr{Int64}, Ptr{Cfloat}, Cfloat), ctx, length(s), s, e, w, p))
    elseif isa(reg, PM.ARDRegularizer) && which == :Y
        s, e = starts(reg.col_ranges), stops(reg.col_ranges)
        a, b = f32(collect(reg.alpha)), f32(collect(reg.beta))
        GC.@preserve s e a b chk(ccall((:pmf_add_yreg_ard, LIB[]), Cint,
            (Ptr{Cvoid}, Cint, Ptr{Int64}, Ptr{Int64}, Ptr{Cfloat}, Ptr{Cfloat}, Cfloat), ctx, length(s), s, e, a, b, p))
    elseif isa(reg, PM.FeatureSetARDReg) && which == :Y
        a, b = f32(reg.alpha), f32(reg.beta)
        GC.@preserve a b chk(ccall((:pmf_add_yreg_fsard, LIB[]), Cint,
            (Ptr{Cvoid}, Ptr{Cfloat}, Ptr{Cfloat}, Cfloat), ctx, a, b, p))
    elseif isa(reg, PM.CompositeRegularizer)
        for (r, q) in zip(reg.regularizers, reg.mixture_p)
            add_reg!(ctx, which, r, p * Float32(q))
        end
    else
        error("PathMatFacHIP: regularizer $(typeof(reg)) is not supported by the HIP path (Network/L1/SelectiveL1 are out of scope)")
    end
end

function marshal!(ctx, mf)
    X, Y = f32(mf.X), f32(mf.Y)
    K = size(X, 1)
    GC.@preserve X Y chk(ccall((:pmf_set_factors, LIB[]), Cint, (Ptr{Cvoid}, Ptr{Cfloat}, Ptr{Cfloat}, Cint), ctx, X, Y, K))
    layers = map(unwrap, mf.col_transform.layers)
    N = size(Y, 2)
    ls = isa(layers[1], PM.ColScale) ? f32(layers[1].logsigma) : zeros(Float32, N)
    mu = isa(layers[3], PM.ColShift) ? f32(layers[3].mu) : zeros(Float32, N)
    GC.@preserve ls mu chk(ccall((:pmf_set_col_params, LIB[]), Cint, (Ptr{Cvoid}, Ptr{Cfloat}, Ptr{Cfloat}), ctx, ls, mu))
    ba = isa(layers[2], PM.BatchScale) ? layers[2].logdelta : (isa(layers[4], PM.BatchShift) ? layers[4].theta : nothing)
    nviews = ba === nothing ? 0 : length(ba.col_ranges)
    chk(ccall((:pmf_set_n_batch_views, LIB[]), Cint, (Ptr{Cvoid}, Cint), ctx, nviews))
    for v in 1:nviews
        cr = ba.col_ranges[v]
        bor = batch_of_row(ba.row_batches[v])
        ld = isa(layers[2], PM.BatchScale) ? f32(layers[2].logdelta.values[v]) : zeros(Float32, size(ba.values[v]))
        th = isa(layers[4], PM.BatchShift) ? f32(layers[4].theta.values[v]) : zeros(Float32, size(ba.values[v]))
        GC.@preserve bor ld th chk(ccall((:pmf_set_batch_view, LIB[]), Cint,
            (Ptr{Cvoid}, Cint, Int64, Int64, Cint, Ptr{Int32}, Ptr{Cfloat}, Ptr{Cfloat}),
            ctx, v - 1, cr.start, cr.stop, size(ld, 1), bor, ld, th))
    end
    nm = mf.noise_model
    s, e = starts(nm.col_ranges), stops(nm.col_ranges)
    kinds = Cint[NOISE_KIND[noise_name(n)] for n in nm.noises]
    w = f32(noise_weights(nm))
    GC.@preserve s e kinds w chk(ccall((:pmf_set_noise, LIB[]), Cint,
        (Ptr{Cvoid}, Cint, Ptr{Int64}, Ptr{Int64}, Ptr{Cint}, Ptr{Cfloat}), ctx, length(s), s, e, kinds, w))
    chk(ccall((:pmf_clear_xreg, LIB[]), Cint, (Ptr{Cvoid},), ctx)); add_reg!(ctx, :X, mf.X_reg)
    chk(ccall((:pmf_clear_yreg, LIB[]), Cint, (Ptr{Cvoid},), ctx)); add_reg!(ctx, :Y, mf.Y_reg)
    marshal_layer_regs!(ctx, mf.col_transform_reg, nviews)
end

function marshal_layer_regs!(ctx, sr, nviews)
    nul = Ptr{Cfloat}(C_NULL)
    if !isa(sr, PM.SequenceReg)
        return chk(ccall((:pmf_set_layer_regs, LIB[]), Cint, (Ptr{Cvoid}, Cint, Ptr{Int64}, Ptr{Int64},
            Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}),
            ctx, 0, C_NULL, C_NULL, nul, nul, nul, nul, nul, nul, nul, nul))
    end
    regs = map(unwrapreg, sr.regs)
    s, e = Int64[], Int64[]
    wls = cls = wmu = cmu = wld = cld = wth = cth = Float32[]
    if isa(regs[1], PM.ColParamReg) && isa(regs[3], PM.ColParamReg)
        s, e = starts(regs[1].col_ranges), stops(regs[1].col_ranges)
        wls, cls = f32(collect(regs[1].weights)), f32(collect(regs[1].centers))
        wmu, cmu = f32(collect(regs[3].weights)), f32(collect(regs[3].centers))
    end
    if nviews > 0 && isa(regs[2], PM.BatchArrayReg) && isa(regs[4], PM.BatchArrayReg)
        wld, cld = f32(vcat(regs[2].weights...)), f32(vcat(regs[2].centers...))
        wth, cth = f32(vcat(regs[4].weights...)), f32(vcat(regs[4].centers...))
    end
    p(a) = isempty(a) ? nul : pointer(a)
    GC.@preserve s e wls cls wmu cmu wld cld wth cth chk(ccall((:pmf_set_layer_regs, LIB[]), Cint,
        (Ptr{Cvoid}, Cint, Ptr{Int64}, Ptr{Int64}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat},
         Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cfloat}),
        ctx, length(s), s, e, p(wls), p(cls), p(wmu), p(cmu), p(wld), p(cld), p(wth), p(cth)))
end

mask(xs, T) = Cint(sum((isa(x, T) ? 1 : 0) << (i - 1) for (i, x) in enumerate(xs)))

"""Replacement for PathMatFac.mf_fit! (src/fit.jl:9-38)."""
function mf_fit!(model::PM.PathMatFacModel; update_X=false, update_Y=false, update_col_layers=false,
                 opt=nothing, lr=0.01, max_epochs=1000, epoch=1, rel_tol=1e-6, abs_tol=1e-9, tol_max_iters=3,
                 verbosity=1, print_iter=10, capacity=10^8, keep_history=true, kwargs...)
    ctx = context!(model)
    mf = model.matfac
    marshal!(ctx, mf)
    opt === nothing && (opt = PM.construct_optimizer(model, lr))
    if get(OPT_KEY, model, UInt(0)) != objectid(opt)     # new optimizer object => fresh AdaGrad state (src/fit.jl:55)
        chk(ccall((:pmf_set_optimizer, LIB[]), Cint, (Ptr{Cvoid}, Cint, Cfloat, Cfloat, Cfloat, Cfloat),
                  ctx, 0, opt.eta, opt.epsilon, 0.9f0, 0.999f0))
        OPT_KEY[model] = objectid(opt)
    else                                                  # same optimizer, possibly halved eta (src/fit.jl:64)
        chk(ccall((:pmf_set_lr, LIB[]), Cint, (Ptr{Cvoid}, Cfloat), ctx, opt.eta))
    end
    frozen = mask(mf.col_transform.layers, PM.FrozenLayer) | mask(mf.col_transform.layers, Function)
    frozen_regs = isa(mf.col_transform_reg, PM.SequenceReg) ? mask(mf.col_transform_reg.regs, PM.FrozenRegularizer) : Cint(0)
    opts = FitOpts(update_X, update_Y, update_col_layers, frozen, frozen_regs, max_epochs, epoch, tol_max_iters,
                   keep_history, verbosity, print_iter, 0, abs_tol, rel_tol, capacity)
    trace = zeros(Cdouble, max(max_epochs - epoch + 1, 1))
    res = FitResult(0, 0, 0, length(trace), 0.0, pointer(trace), 0.0)
    GC.@preserve trace chk(ccall((:pmf_fit, LIB[]), Cint, (Ptr{Cvoid}, Ref{FitOpts}, Ref{FitResult}), ctx, opts, res))
    unmarshal!(ctx, mf, update_X, update_Y, update_col_layers)
    return Dict("term_code" => TERM_CODES[res.term_code + 1], "epochs" => Int(res.epochs),
                "loss" => trace[1:res.n_trace], "total_loss" => res.final_loss)
end

function unmarshal!(ctx, mf, update_X, update_Y, update_col_layers)
    if update_X || update_Y
        X, Y = similar(mf.X, Float32), similar(mf.Y, Float32)
        GC.@preserve X Y chk(ccall((:pmf_get_factors, LIB[]), Cint, (Ptr{Cvoid}, Ptr{Cfloat}, Ptr{Cfloat}), ctx, X, Y))
        update_X && (mf.X .= X)
        update_Y && (mf.Y .= Y)
    end
    if update_col_layers
        layers = map(unwrap, mf.col_transform.layers)
        N = size(mf.Y, 2)
        ls, mu = zeros(Float32, N), zeros(Float32, N)
        GC.@preserve ls mu chk(ccall((:pmf_get_col_params, LIB[]), Cint, (Ptr{Cvoid}, Ptr{Cfloat}, Ptr{Cfloat}), ctx, ls, mu))
        isa(layers[1], PM.ColScale) && (layers[1].logsigma .= ls)
        isa(layers[3], PM.ColShift) && (layers[3].mu .= mu)
        if isa(layers[4], PM.BatchShift)
            for v in 1:length(layers[4].theta.values)
                ld, th = similar(layers[2].logdelta.values[v], Float32), similar(layers[4].theta.values[v], Float32)
                GC.@preserve ld th chk(ccall((:pmf_get_batch_view, LIB[]), Cint,
                                             (Ptr{Cvoid}, Cint, Ptr{Cfloat}, Ptr{Cfloat}), ctx, v - 1, ld, th))
                layers[2].logdelta.values[v] .= ld
                layers[4].theta.values[v] .= th
            end
        end
    end
end

"""Point PathMatFac's drop-in boundary at the HIP library."""
function install!(libpath::AbstractString="libpmf_hip.so")
    LIB[] = libpath
    @eval PathMatFac mf_fit!(model::PathMatFacModel; kwargs...) = Main.PathMatFacHIP.mf_fit!(model; kwargs...)
    return nothing
end

function release!(model)
    haskey(CTX, model) && (ccall((:pmf_destroy, LIB[]), Cint, (Ptr{Cvoid},), CTX[model]); delete!(CTX, model))
    delete!(DATA_KEY, model); delete!(OPT_KEY, model)
    return nothing
end

end # module
