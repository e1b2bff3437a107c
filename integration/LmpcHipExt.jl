# Julia glue for liblmpc_hip.so -- the file INTEGRATION.md walks through.  Written against include/lmpc_hip.h;
# Julia is not installed in the build image, so it has not been executed there (examples/abi_check.c is the
# C client that is built and run by the test-suite).
module LmpcHipExt
using LinearMPC, LinearAlgebra
import DAQP
const liblmpc = get(ENV, "LMPC_HIP_LIB", "liblmpc_hip.so")

struct LmpcSettings            # == lmpc_settings (include/lmpc_hip.h)
    primal_tol::Cdouble; dual_tol::Cdouble; zero_tol::Cdouble; progress_tol::Cdouble
    fval_bound::Cdouble; rho_soft::Cdouble; cycle_tol::Cint; iter_limit::Cint
    eps_prox::Cdouble; eta_prox::Cdouble
end
# The solver settings the user has put on mpc.opt_model with DAQP.settings(mpc.opt_model, Dict(...))
# (/root/reference/docs/src/manual/solver.md:19-22) are read back from the DAQP model, not assumed: DAQP.settings(model)
# returns the C struct DAQPSettings with these field names.  rho_soft is what setup! itself wrote there,
# 1 / mpc.settings.soft_weight (/root/reference/src/setup.jl:26).  eps_prox / eta_prox select the backend's
# proximal-point mode (a semidefinite H); pivot_tol and the B&B tolerances have no counterpart.
function LmpcSettings(mpc::LinearMPC.MPC)
    d = DAQP.settings(mpc.opt_model)
    return LmpcSettings(d.primal_tol, d.dual_tol, d.zero_tol, d.progress_tol, d.fval_bound, d.rho_soft,
                        d.cycle_tol, d.iter_limit, d.eps_prox, d.eta_prox)
end

mutable struct BatchedModel    # stands next to mpc.opt_model
    h::Ptr{Cvoid}; n::Int; nout::Int; nth::Int; words::Int
    mpqp::Any                  # the mpQP OBJECT the handle was built from (setup! always makes a new one, setup.jl:9)
    settings::LmpcSettings     # the solver settings the handle was last given
end
free!(bm::BatchedModel) = (bm.h == C_NULL || ccall((:lmpc_free, liblmpc), Cvoid, (Ptr{Cvoid},), bm.h); bm.h = C_NULL; nothing)

"setup!(mpc) for the batched backend: same inputs DAQP.setup gets (setup.jl:11-13)"
function setup_batched(mpc::LinearMPC.MPC; nout=mpc.model.nu, device=0)
    mpc.mpqp_issetup || LinearMPC.setup!(mpc)
    q = mpc.mpQP
    # setup.jl:11-13: DAQP.setup(model, H, f, A, bu, bl, senses; break_points = mpQP.break_points,
    # is_avi = !mpQP.is_symmetric) -- lmpc_setup_ex takes the same two keywords.  A non-symmetric H (several
    # objectives) sets the handle up for the variational inequality; a non-empty break_points (prioritised
    # constraints) is answered with LMPC_ERR_UNSUPPORTED (-103) by the library, never ignored.
    n = size(q.H,1); m = length(q.bu); ms = m - size(q.A,1); nth = size(q.f_theta,2)
    h = Ref{Ptr{Cvoid}}(C_NULL); st = LmpcSettings(mpc); s = Ref(st)
    K = iszero(mpc.K) ? C_NULL : Matrix{Float64}(mpc.K[1:nout, :])
    bp = Vector{Cint}(q.break_points)
    flag = ccall((:lmpc_setup_ex, liblmpc), Cint,
        (Ref{Ptr{Cvoid}}, Cint,Cint,Cint,Cint,Cint, Ptr{Cdouble},Ptr{Cdouble},Ptr{Cdouble},Ptr{Cdouble},
         Ptr{Cdouble},Ptr{Cdouble},Ptr{Cdouble},Ptr{Cint},Ptr{Cdouble},Cint,Ref{LmpcSettings},Ptr{Cint},Cint,Cint,Cint),
        h, n,m,ms,nth,nout, q.H,q.f,q.f_theta,q.A,q.bu,q.bl,q.W,q.senses, K, mpc.model.nx, s,
        isempty(bp) ? C_NULL : bp, length(bp), !q.is_symmetric, device)
    flag == 1 || error("lmpc_setup_ex failed ($flag): ", unsafe_string(ccall((:lmpc_last_error, liblmpc), Cstring, (Ptr{Cvoid},), C_NULL)))
    words = ccall((:lmpc_active_words, liblmpc), Cint, (Ptr{Cvoid},), h[])
    bm = BatchedModel(h[], n, nout, nth, words, q, st)
    finalizer(free!, bm)
    return bm
end

"solve(mpc, Θ) for a matrix of parameters (nth × N): the batched twin of utils.jl:268-283"
function LinearMPC.solve(bm::BatchedModel, Θ::Matrix{Float64})
    @assert size(Θ,1) == bm.nth
    N = size(Θ,2); X = Matrix{Float64}(undef, bm.nout, N); flags = Vector{Cint}(undef, N)
    rc = ccall((:lmpc_solve_batch, liblmpc), Cint,
        (Ptr{Cvoid}, Int64, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cint}, Ptr{Cint}, Ptr{UInt64}, Ptr{UInt64}),
        bm.h, N, Θ, X, flags, C_NULL, C_NULL, C_NULL)
    rc == 1 || error("lmpc_solve_batch failed ($rc)")
    return X, flags
end

# ---- the literal drop-in: solve(mpc, θ) for ONE parameter vector, the call compute_control / Simulation make
# (/root/reference/src/utils.jl:43-51, :268-283; /root/reference/src/simulation.jl:106).  A method on (MPC, AbstractVector)
# is more specific than the package's own solve(mpc::MPC, θ), so after `using LmpcHipExt` the unchanged
# compute_control(mpc, x), compute_control_trajectory and Simulation(mpc; ...) run on lmpc_solve_one.
# Returns what DAQP.solve returns: (x*, fval, exitflag, info); fval = 1/2 x'Hx + (f + f_θ θ)'x is formed on the host
# from x* (the reference reads x* and exitflag only, utils.jl:45-48).  One handle per MPC object, with nout = n
# (compute_control subtracts K·x itself, utils.jl:48-49) and the user's DAQP settings.
# The cache: one handle per live MPC object (WeakKeyDict: an MPC that is garbage collected takes its handle along,
# the BatchedModel's finalizer frees it).  A handle is STALE as soon as the MPC holds another mpQP object than the one
# it was built from: every set_*! only clears mpc.mpqp_issetup (setup.jl:36-160) and the next setup! -- called by
# solve (utils.jl:269), by us, or by the user -- builds a NEW mpQP of possibly the same dimensions (setup.jl:9), so
# dimensions say nothing; object identity does.  The solver settings on mpc.opt_model are re-read on every solve
# (a struct of ten numbers: lmpc_settings) and pushed with lmpc_set_settings when they differ from what the handle was last
# given, so DAQP.settings(mpc.opt_model, Dict(...)) after the first solve reaches the GPU path too.
# (The Python mirror linearmpc.jl_amd/mpc.py::MPC._model_for is this logic line for line; the test-suite runs the
# failing sequences on it: tests/test_gpu_parity.py::test_solve_mpc_theta_drop_in_and_user_settings.)
const _models = WeakKeyDict{Any,BatchedModel}()
function _model_for(mpc::LinearMPC.MPC)
    mpc.mpqp_issetup || LinearMPC.setup!(mpc)
    mpc.mpqp_issetup || throw("Could not setup optimization problem")      # as utils.jl:270
    bm = get(_models, mpc, nothing)
    if bm === nothing || bm.mpqp !== mpc.mpQP || LmpcSettings(mpc).eps_prox != bm.settings.eps_prox
        # (eps_prox is part of the factorisation: a handle set up with another value is as stale as one of another mpQP)
        bm === nothing || free!(bm)                     # the stale handle goes now, not at some later GC
        K = mpc.K; mpc.K = zero(K)                      # nout = n handle without the feedback folded in
        try bm = setup_batched(mpc; nout=size(mpc.mpQP.H,1)) finally mpc.K = K end
        _models[mpc] = bm
    end
    st = LmpcSettings(mpc)
    if st != bm.settings
        rc = ccall((:lmpc_set_settings, liblmpc), Cint, (Ptr{Cvoid}, Ref{LmpcSettings}), bm.h, Ref(st))
        rc == 1 || error("lmpc_set_settings failed ($rc)")
        bm.settings = st
    end
    return bm
end
function LinearMPC.solve(mpc::LinearMPC.MPC, θ::AbstractVector{<:Real})
    bm = _model_for(mpc)
    th = Vector{Float64}(θ); x = Vector{Float64}(undef, bm.nout)
    flag = ccall((:lmpc_solve_one, liblmpc), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}), bm.h, th, x)
    flag <= -100 && error("lmpc_solve_one failed ($flag): ",
                          unsafe_string(ccall((:lmpc_last_error, liblmpc), Cstring, (Ptr{Cvoid},), bm.h)))
    q = mpc.mpQP
    fval = 0.5*dot(x, q.H, x) + dot(q.f .+ q.f_theta*th, x)
    return x, fval, Int(flag), (status = flag >= 1 ? :Solved : :Failed, exitflag = Int(flag))
end
"several parameter batches already resident on the GPU (device pointers, e.g. from AMDGPU.jl arrays), ONE call: on the
handles the one-launch kernel covers the batches share one kernel launch (include/lmpc_hip.h, lmpc_solve_batches_device)"
function solve_batches_device!(bm::BatchedModel, N::Integer, thetas::Vector{Ptr{Cdouble}}, xs::Vector{Ptr{Cdouble}},
                               flags::Vector{Ptr{Cint}}; stream::Ptr{Cvoid}=C_NULL)
    length(thetas) == length(xs) == length(flags) || error("one x and one flag array per parameter batch")
    rc = ccall((:lmpc_solve_batches_device, liblmpc), Cint,
        (Ptr{Cvoid}, Cint, Int64, Ptr{Ptr{Cdouble}}, Ptr{Ptr{Cdouble}}, Ptr{Ptr{Cint}}, Ptr{Cvoid}),
        bm.h, length(thetas), N, thetas, xs, flags, stream)
    rc == 1 || error("lmpc_solve_batches_device failed ($rc)")
    return nothing
end

"drop (and free) the handle of an MPC now; never needed for correctness -- _model_for notices a new mpQP by itself"
reset_batched!(mpc::LinearMPC.MPC) = (bm = pop!(_models, mpc, nothing); bm === nothing || free!(bm); nothing)

"single-precision twin (the reference's codegen float_type=\"float\" build of the same path)"
function solve_f32(bm::BatchedModel, Θ::Matrix{Float32})
    N = size(Θ,2); X = Matrix{Float32}(undef, bm.nout, N); flags = Vector{Cint}(undef, N)
    rc = ccall((:lmpc_solve_batch_f32, liblmpc), Cint,
        (Ptr{Cvoid}, Int64, Ptr{Cfloat}, Ptr{Cfloat}, Ptr{Cint}, Ptr{Cint}, Ptr{UInt64}, Ptr{UInt64}),
        bm.h, N, Θ, X, flags, C_NULL, C_NULL, C_NULL)
    rc == 1 || error("lmpc_solve_batch_f32 failed ($rc)")
    return X, flags
end

"the generated controller's signature, batched: control (nu × N) in = previous control, out = u*"
function mpc_compute_control!(bm::BatchedModel, control::Matrix{Float64}, state::Matrix{Float64},
                              reference=C_NULL, disturbance=C_NULL, parameter=C_NULL; warm=false)
    N = size(control,2); flags = Vector{Cint}(undef, N)
    rc = ccall((:lmpc_compute_control, liblmpc), Cint,
        (Ptr{Cvoid}, Int64, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cint}, Cint),
        bm.h, N, control, state, reference, disturbance, parameter, flags, warm)
    rc == 1 || error("lmpc_compute_control failed ($rc)")
    return flags
end
# once per handle, from the MPC's parameter dimensions (get_parameter_dims, mpc2mpqp.jl:147-164):
#   lay = LmpcParamLayout(nx, nr, nd, nuprev, np, cond ? mpc.Np : 0, cond ? pointer(mpc.traj2setpoint) : C_NULL)
#   ccall((:lmpc_set_parameter_layout, liblmpc), Cint, (Ptr{Cvoid}, Ref{LmpcParamLayout}), bm.h, lay)

"compute_control for N scenarios at once (utils.jl:43-51 without the mutable mpc.uprev)"
function compute_control_batch(mpc, bm::BatchedModel, X0::Matrix; R=nothing, Uprev=nothing, check=true)
    N = size(X0,2)
    Θ = reduce(hcat, [LinearMPC.form_parameter(mpc, X0[:,i], isnothing(R) ? nothing : R[:,i], nothing,
                                              isnothing(Uprev) ? zeros(mpc.model.nu) : Uprev[:,i]) for i in 1:N])
    U, flags = LinearMPC.solve(bm, Θ)          # K·x already subtracted inside the library
    check && @assert all(flags .>= 1)
    return U, flags
end

# ---- every GPU of the node behind ONE call (one process, one Θ -- the shape of utils.jl:268-283)
mutable struct MultiModel
    hm::Ptr{Cvoid}; nout::Int; nth::Int
end
function setup_batched_multi(mpc::LinearMPC.MPC; nout=mpc.model.nu, devices=Cint[])    # empty = all visible GPUs
    mpc.mpqp_issetup || LinearMPC.setup!(mpc)
    q = mpc.mpQP
    # (lmpc_setup_multi has no keyword form: it decides is_avi from H like lmpc_setup; priorities are refused here)
    isempty(q.break_points) || error("lmpc: prioritised constraints are not supported")
    n = size(q.H,1); m = length(q.bu); ms = m - size(q.A,1); nth = size(q.f_theta,2)
    hm = Ref{Ptr{Cvoid}}(C_NULL); s = Ref(LmpcSettings(mpc))
    K = iszero(mpc.K) ? C_NULL : Matrix{Float64}(mpc.K[1:nout, :])
    flag = ccall((:lmpc_setup_multi, liblmpc), Cint,
        (Ref{Ptr{Cvoid}}, Cint,Cint,Cint,Cint,Cint, Ptr{Cdouble},Ptr{Cdouble},Ptr{Cdouble},Ptr{Cdouble},
         Ptr{Cdouble},Ptr{Cdouble},Ptr{Cdouble},Ptr{Cint},Ptr{Cdouble},Cint,Ref{LmpcSettings},Ptr{Cint},Cint),
        hm, n,m,ms,nth,nout, q.H,q.f,q.f_theta,q.A,q.bu,q.bl,q.W,q.senses, K, mpc.model.nx, s,
        isempty(devices) ? C_NULL : devices, length(devices))
    flag == 1 || error("lmpc_setup_multi failed ($flag)")
    mm = MultiModel(hm[], nout, nth)
    finalizer(b -> ccall((:lmpc_free_multi, liblmpc), Cvoid, (Ptr{Cvoid},), b.hm), mm)
    return mm
end
"solve(mpc, Θ) across the GPUs: contiguous shards, results straight into X / flags"
function LinearMPC.solve(mm::MultiModel, Θ::Matrix{Float64}, X::Matrix{Float64}, flags::Vector{Cint})
    N = size(Θ,2)
    rc = ccall((:lmpc_solve_batch_multi, liblmpc), Cint,
        (Ptr{Cvoid}, Int64, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cint}, Ptr{Cint}, Ptr{UInt64}, Ptr{UInt64}),
        mm.hm, N, Θ, X, flags, C_NULL, C_NULL, C_NULL)
    rc == 1 || error("lmpc_solve_batch_multi failed ($rc)")
    return X, flags
end
end
