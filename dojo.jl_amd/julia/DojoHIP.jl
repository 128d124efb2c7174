# DojoHIP.jl -- thin Julia shim over libdojo_hip.so (include/dojo_hip.h).
#
# NOT EXECUTED IN THE BUILD CONTAINER (no Julia toolchain there); kept deliberately thin: it only
# (a) walks a live Dojo `Mechanism` and fills the C-POD topology, (b) owns a handle, (c) forwards
# step! / simulate! / get_maximal_gradients! for a batch, and (d) installs, on enable!(mechanism), a
# Dojo.mehrotra! method that round-trips that Mechanism through the library (B = 1, body impulses
# JF2/Jτ2 in, solution + μ out) and writes everything mehrotra! mutates back, so that DojoEnvironments
# works unmodified.  The exact call sequence of (d) is replayed from C by tests/c_driver/shim_driver.c.  The same calls are exercised from
# Python by dojo.jl_amd/host/dojo_amd/api.py, which the parity tests drive.
module DojoHIP

using Dojo
using StaticArrays
using Libdl

# The library is opened when the module is LOADED (not when it is precompiled): __init__ puts GPU_MAX_HW_QUEUES into the
# environment before the HIP runtime starts (one hardware queue per environment group of dojo_step_dev / dojo_rollout,
# INTEGRATION.md "Streams and hardware queues"), resolves DOJO_HIP_LIB and dlopens it; every @ccall below goes through a
# dlsym'ed function pointer.
const LIBH = Ref{Ptr{Cvoid}}(C_NULL)
const SYMS = Dict{Symbol,Ptr{Cvoid}}()
function __init__()
    get!(ENV, "GPU_MAX_HW_QUEUES", "24")
    LIBH[] = Libdl.dlopen(get(ENV, "DOJO_HIP_LIB", joinpath(@__DIR__, "..", "csrc", "libdojo_hip.so")))
    empty!(SYMS)
end
fn(name::Symbol) = get!(() -> Libdl.dlsym(LIBH[], name), SYMS, name)

# ---- C PODs (layout identical to include/dojo_hip.h) -------------------------------------------
struct CBody;      mass::Cdouble; inertia::NTuple{9,Cdouble}; end
struct CJointHalf
    nl::Int32; nlim::Int32
    cmask::NTuple{9,Cdouble}; amask::NTuple{9,Cdouble}
    spring::Cdouble; damper::Cdouble
    spring_offset::NTuple{3,Cdouble}; limit_lo::NTuple{3,Cdouble}; limit_hi::NTuple{3,Cdouble}
end
struct CJoint
    parent::Int32; child::Int32; spring_on::Int32; damper_on::Int32
    vertex_parent::NTuple{3,Cdouble}; vertex_child::NTuple{3,Cdouble}; orientation_offset::NTuple{4,Cdouble}
    tra::CJointHalf; rot::CJointHalf
end
struct CContact
    body::Int32; model::Int32; friction_coefficient::Cdouble
    normal::NTuple{3,Cdouble}; tangent::NTuple{6,Cdouble}; origin::NTuple{3,Cdouble}; radius::Cdouble; offset::NTuple{3,Cdouble}
    collision::Int32; child_body::Int32; child_origin::NTuple{3,Cdouble}; child_radius::Cdouble      # SphereSphereCollision (collision = 1)
end
struct CTopology
    n_bodies::Int32; n_joints::Int32; n_contacts::Int32; reserved::Int32
    timestep::Cdouble; input_scaling::Cdouble; gravity::NTuple{3,Cdouble}
    bodies::Ptr{CBody}; joints::Ptr{CJoint}; contacts::Ptr{CContact}
end
struct CSolverOptions
    rtol::Cdouble; btol::Cdouble; undercut::Cdouble; no_progress_undercut::Cdouble
    max_iter::Int32; max_ls::Int32; no_progress_max::Int32; reserved::Int32
end

pad(v, n) = ntuple(i -> i <= length(v) ? Float64(v[i]) : 0.0, n)
rowmajor(M) = pad(vec(permutedims(Matrix(M))), 9)          # k x 3 mask -> 9 doubles, row-major, zero padded

function half(j::Dojo.Joint{T,Nλ,Nb,N,Nb½}) where {T,Nλ,Nb,N,Nb½}
    lo = Nb½ > 0 ? j.joint_limits[1] : Float64[]
    hi = Nb½ > 0 ? j.joint_limits[2] : Float64[]
    CJointHalf(Nλ, Nb½, rowmajor(Dojo.constraint_mask(j)), rowmajor(Dojo.nullspace_mask(j)),
               j.spring, j.damper, pad(j.spring_offset, 3), pad(lo, 3), pad(hi, 3))
end

"walk a Mechanism (bodies / joints / contacts in their live order) -> C arrays"
function export_topology(m::Dojo.Mechanism{T,Nn,Ne,Nb,Ni}) where {T,Nn,Ne,Nb,Ni}
    bidx(id) = id == 0 ? Int32(-1) : Int32(id - Ne - 1)      # node id -> 0-based index into mechanism.bodies
    bodies = [CBody(b.mass, pad(vec(permutedims(Matrix(b.inertia))), 9)) for b in m.bodies]
    joints = [CJoint(bidx(j.parent_id), bidx(j.child_id), Int32(j.spring), Int32(j.damper),
                     pad(j.translational.vertices[1], 3), pad(j.translational.vertices[2], 3),
                     pad(Dojo.vector(j.rotational.orientation_offset), 4), half(j.translational), half(j.rotational)) for j in m.joints]
    contacts = CContact[]
    for c in m.contacts
        (c.model isa Dojo.NonlinearContact || c.model isa Dojo.ImpactContact || c.model isa Dojo.LinearContact) || error("DojoHIP: unknown contact model")
        impact = c.model isa Dojo.ImpactContact
        col = c.model.collision
        if col isa Dojo.SphereSphereCollision
            # body-body contact (src/contacts/collisions/sphere_sphere.jl): forward only.  Where the child body's joint hangs on the parent body the
            # contact is an edge of the tree (the fast quad builds); between any other two bodies the library carries it as a cut element (general
            # lane-mapping builds, at most two per mechanism, NonlinearContact / ImpactContact).  A child body WITHOUT a joint (test/collisions.jl:2-58)
            # gets a Floating joint to the parent here -- the tree-edge form: no rows; its six inputs come last in the library's u and must stay zero
            # (pad u accordingly).
            cb = bidx(c.child_id)
            if !any(j -> j.child == cb, joints)
                free = CJointHalf(0, 0, ntuple(_ -> 0.0, 9), rowmajor([1.0 0 0; 0 1 0; 0 0 1]), 0.0, 0.0, ntuple(_ -> 0.0, 3), ntuple(_ -> 0.0, 3), ntuple(_ -> 0.0, 3))
                push!(joints, CJoint(bidx(c.parent_id), cb, Int32(0), Int32(0), ntuple(_ -> 0.0, 3), ntuple(_ -> 0.0, 3), (1.0, 0.0, 0.0, 0.0), free, free))
            end
            push!(contacts, CContact(bidx(c.parent_id), impact ? 1 : (c.model isa Dojo.LinearContact ? 2 : 0), impact ? 0.0 : c.model.friction_coefficient, ntuple(_ -> 0.0, 3), ntuple(_ -> 0.0, 6),
                                     pad(col.origin_parent, 3), col.radius_parent, ntuple(_ -> 0.0, 3), Int32(1), cb, pad(col.origin_child, 3), col.radius_child))
            continue
        end
        col isa Dojo.SphereHalfSpaceCollision || error("DojoHIP: only SphereHalfSpaceCollision and SphereSphereCollision are supported")
        # (LinearContact: the library has the reference's friction_parameterization [0 1; 0 -1; 1 0; -1 0] built in, src/contacts/linear.jl:33-38)
        push!(contacts, CContact(bidx(c.parent_id), impact ? 1 : (c.model isa Dojo.LinearContact ? 2 : 0), impact ? 0.0 : c.model.friction_coefficient, pad(col.contact_normal', 3),
                                 impact ? ntuple(_ -> 0.0, 6) : pad(vec(permutedims(Matrix(col.contact_tangent))), 6), pad(col.contact_origin, 3),
                                 col.contact_radius, pad(col.contact_offset, 3), Int32(0), Int32(-1), ntuple(_ -> 0.0, 3), 0.0))
    end
    return bodies, joints, contacts
end

mutable struct BatchedMechanism{T}
    handle::Ptr{Cvoid}
    mechanism::Dojo.Mechanism
    batch::Int
    nz::Int; nx::Int; nu::Int
end

check(rc) = rc == 0 || error("libdojo_hip: " * unsafe_string(@ccall $(fn(:dojo_last_error))()::Cstring))
"the text of the last failure on this handle (per handle; dojo_last_error is process-wide)"
last_error(bm) = unsafe_string(@ccall $(fn(:dojo_handle_error))(bm.handle::Ptr{Cvoid})::Cstring)

function BatchedMechanism(m::Dojo.Mechanism, batch::Int; T=Float32, device::Int=0)
    bodies, joints, contacts = export_topology(m)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve bodies joints contacts begin
        topo = CTopology(length(bodies), length(joints), length(contacts), 0, m.timestep, m.input_scaling,
                         pad(m.gravity, 3), pointer(bodies), pointer(joints), pointer(contacts))
        check(@ccall $(fn(:dojo_create))(Ref(topo)::Ref{CTopology}, batch::Int32, (T == Float32 ? 1 : 0)::Int32, device::Int32, h::Ref{Ptr{Cvoid}})::Cint)
    end
    bm = BatchedMechanism{T}(h[], m, batch, 13length(m.bodies), 12length(m.bodies), Dojo.input_dimension(m))
    finalizer(b -> (@ccall $(fn(:dojo_destroy))(b.handle::Ptr{Cvoid})::Cvoid), bm)
    return bm
end

function set_options!(bm::BatchedMechanism, o::Dojo.SolverOptions)
    c = CSolverOptions(o.rtol, o.btol, o.undercut, o.no_progress_undercut, o.max_iter, o.max_ls, o.no_progress_max, 0)
    check(@ccall $(fn(:dojo_set_options))(bm.handle::Ptr{Cvoid}, Ref(c)::Ref{CSolverOptions})::Cint)
end

# z: nz x B and u: nu x B column-major Julia matrices == the row-major [B, nz] / [B, nu] of the ABI
"step!(mechanism, z, u; opts): batched, returns (z_next, status)"
function Dojo.step!(bm::BatchedMechanism{T}, z::Matrix{T}, u::Matrix{T}; opts=Dojo.SolverOptions{Float64}(), with_gradient::Bool=false) where T
    set_options!(bm, opts)
    zn = similar(z); status = Vector{Int32}(undef, bm.batch); iters = Vector{Int32}(undef, bm.batch)
    check(@ccall $(fn(:dojo_step))(bm.handle::Ptr{Cvoid}, z::Ptr{T}, u::Ptr{T}, zn::Ptr{T}, status::Ptr{Int32}, iters::Ptr{Int32}, with_gradient::Int32)::Cint)
    return zn, status
end

"""
The vector `Dojo.step!(mechanism, z, u)` literally returns is `get_next_state` AFTER `update_state!` (src/simulation/step.jl:28),
i.e. the mechanism's next state advanced once more; `step!(bm, ...)` above returns the mechanism's next state itself (what the
next step and `get_state` consume).  `reference_return(bm, zn)` maps the one to the other.
"""
function reference_return(bm::BatchedMechanism{T}, zn::Matrix{T}) where T
    zr = similar(zn)
    check(@ccall $(fn(:dojo_next_state))(bm.handle::Ptr{Cvoid}, zn::Ptr{T}, zr::Ptr{T})::Cint)
    return zr
end

"get_maximal_gradients!(mechanism, z, u; opts): batched -> (jacobian_state[12Nb,12Nb,B], jacobian_control[12Nb,nu,B])"
function Dojo.get_maximal_gradients!(bm::BatchedMechanism{T}, z::Matrix{T}, u::Matrix{T}; opts=Dojo.SolverOptions{Float64}()) where T
    Dojo.step!(bm, z, u; opts, with_gradient=true)
    dz = Array{T}(undef, bm.nx, bm.nx, bm.batch); du = Array{T}(undef, bm.nu, bm.nx, bm.batch)      # ABI is row-major [B,nx,nx]/[B,nx,nu]
    check(@ccall $(fn(:dojo_gradients))(bm.handle::Ptr{Cvoid}, dz::Ptr{T}, du::Ptr{T})::Cint)
    return permutedims(dz, (2, 1, 3)), permutedims(du, (2, 1, 3))
end

"simulate! with pre-sampled controls U[nu, B, H] -> Z[nz, B, H], status[B, H]"
function Dojo.simulate!(bm::BatchedMechanism{T}, z0::Matrix{T}, U::Array{T,3}; opts=Dojo.SolverOptions{Float64}()) where T
    set_options!(bm, opts)
    H = size(U, 3)
    Z = Array{T}(undef, bm.nz, bm.batch, H); status = Matrix{Int32}(undef, bm.batch, H)
    check(@ccall $(fn(:dojo_rollout))(bm.handle::Ptr{Cvoid}, z0::Ptr{T}, U::Ptr{T}, H::Int32, Z::Ptr{T}, status::Ptr{Int32})::Cint)
    return Z, status
end

"which states the IFT data blocks are evaluated at: 0 = as the reference does after step! (post-update_state!), 1 = at the solved step (consistent)"
set_gradient_mode!(bm::BatchedMechanism, mode::Integer) = check(@ccall $(fn(:dojo_set_gradient_mode))(bm.handle::Ptr{Cvoid}, Int32(mode)::Int32)::Cint)

"external forces for every body of every environment: fext[6, Nb, B] = [state.Fext (world); state.τext (body frame)]; `nothing` removes them"
function set_external_force!(bm::BatchedMechanism{T}, fext::Union{Nothing,Array{T,3}}) where T
    check(@ccall $(fn(:dojo_set_external_force))(bm.handle::Ptr{Cvoid}, (fext === nothing ? C_NULL : pointer(fext))::Ptr{T})::Cint)
end

"simulate!(...; record=true): as above plus the Storage rows [25, Nb, B, H] (x q v ω px pq vl ωl, storage.jl:50-67)"
function simulate_storage!(bm::BatchedMechanism{T}, z0::Matrix{T}, U::Array{T,3}; opts=Dojo.SolverOptions{Float64}()) where T
    set_options!(bm, opts)
    H = size(U, 3); Nb = length(bm.mechanism.bodies)
    Z = Array{T}(undef, bm.nz, bm.batch, H); S = Array{T}(undef, 25, Nb, bm.batch, H); status = Matrix{Int32}(undef, bm.batch, H)
    check(@ccall $(fn(:dojo_simulate))(bm.handle::Ptr{Cvoid}, z0::Ptr{T}, U::Ptr{T}, H::Int32, Z::Ptr{T}, S::Ptr{T}, status::Ptr{Int32})::Cint)
    return Z, S, status
end

"get_state(environment): minimal state (+ clamped contact normal impulses, as get_state(::AntARS)) of the last step, [2nu (+Nc), B]"
function get_state(bm::BatchedMechanism{T}; contact_forces::Bool=false) where T
    n = 2 * bm.nu + (contact_forces ? length(bm.mechanism.contacts) : 0)
    obs = Matrix{T}(undef, n, bm.batch)
    check(@ccall $(fn(:dojo_observe))(bm.handle::Ptr{Cvoid}, obs::Ptr{T}, Int32(contact_forces)::Int32)::Cint)
    return obs
end

"get_contact_gradients(mechanism) at the solution of the last get_maximal_gradients!: jacobian_contact[12Nb, 5Nc, B]"
function get_contact_gradients!(bm::BatchedMechanism{T}, ncontacts::Int) where T
    dc = Array{T}(undef, 5 * ncontacts, bm.nx, bm.batch)                      # ABI is row-major [B, nx, 5Nc]
    check(@ccall $(fn(:dojo_contact_gradients))(bm.handle::Ptr{Cvoid}, dc::Ptr{T})::Cint)
    return permutedims(dc, (2, 1, 3))
end

"minimal_to_maximal(mechanism, x): batched, x is 2nu x B (per joint [Δx; Δθ; Δv; Δω]) -> z (13Nb x B)"
function Dojo.minimal_to_maximal(bm::BatchedMechanism{T}, x::Matrix{T}) where T
    z = Matrix{T}(undef, bm.nz, bm.batch)
    check(@ccall $(fn(:dojo_minimal_to_maximal))(bm.handle::Ptr{Cvoid}, x::Ptr{T}, z::Ptr{T})::Cint)
    return z
end

"maximal_to_minimal(mechanism, z): batched"
function Dojo.maximal_to_minimal(bm::BatchedMechanism{T}, z::Matrix{T}) where T
    x = Matrix{T}(undef, 2 * bm.nu, bm.batch)
    check(@ccall $(fn(:dojo_maximal_to_minimal))(bm.handle::Ptr{Cvoid}, z::Ptr{T}, x::Ptr{T})::Cint)
    return x
end

"step_minimal_coordinates!(mechanism, x, u; opts): batched, returns (x_next, status)"
function Dojo.step_minimal_coordinates!(bm::BatchedMechanism{T}, x::Matrix{T}, u::Matrix{T}; opts=Dojo.SolverOptions{Float64}()) where T
    set_options!(bm, opts)
    xn = similar(x); status = Vector{Int32}(undef, bm.batch); iters = Vector{Int32}(undef, bm.batch)
    check(@ccall $(fn(:dojo_step_minimal))(bm.handle::Ptr{Cvoid}, x::Ptr{T}, u::Ptr{T}, xn::Ptr{T}, status::Ptr{Int32}, iters::Ptr{Int32})::Cint)
    return xn, status
end

"get_minimal_gradients!(mechanism, x, u; opts): batched -> (jacobian_state[2nu, 2nu, B], jacobian_control[2nu, nu, B]); also returns x_next"
function Dojo.get_minimal_gradients!(bm::BatchedMechanism{T}, x::Matrix{T}, u::Matrix{T}; opts=Dojo.SolverOptions{Float64}()) where T
    set_options!(bm, opts)
    nm = 2 * bm.nu
    xn = similar(x); status = Vector{Int32}(undef, bm.batch); iters = Vector{Int32}(undef, bm.batch)
    jx = Array{T}(undef, nm, nm, bm.batch); ju = Array{T}(undef, bm.nu, nm, bm.batch)         # ABI is row-major [B, 2nu, 2nu] / [B, 2nu, nu]
    check(@ccall $(fn(:dojo_minimal_gradients))(bm.handle::Ptr{Cvoid}, x::Ptr{T}, u::Ptr{T}, xn::Ptr{T}, status::Ptr{Int32}, iters::Ptr{Int32}, jx::Ptr{T}, ju::Ptr{T})::Cint)
    return permutedims(jx, (2, 1, 3)), permutedims(ju, (2, 1, 3)), xn, status
end

# ---- device-resident batches, scheduling, multi-GPU ------------------------------------------------
# Everything below takes raw device pointers (Ptr{Cvoid}: `pointer(::ROCArray)` of AMDGPU.jl, or whatever allocator owns
# the handle's GPU) and a hipStream_t as Ptr{Cvoid} (C_NULL = the null stream); nothing is synchronized (dojo_hip.h,
# "device-pointer variants").  A training loop keeps z, u, dz, du on the device and calls step_dev! per step.
const DevPtr = Ptr{Cvoid}

"number of GPUs the library sees"
device_count() = Int(@ccall $(fn(:dojo_device_count))()::Cint)

"one differentiable step on device buffers: z [nz, B], u [nu, B] -> z_next, status [B], iters [B], dz [nx, nx, B], du [nx, nu, B] (dz = du = C_NULL: forward only)"
function step_dev!(bm::BatchedMechanism, z::DevPtr, u::DevPtr, z_next::DevPtr, status::DevPtr, iters::DevPtr,
                   dz::DevPtr=C_NULL, du::DevPtr=C_NULL; stream::DevPtr=C_NULL)
    check(@ccall $(fn(:dojo_step_dev))(bm.handle::Ptr{Cvoid}, z::Ptr{Cvoid}, u::Ptr{Cvoid}, z_next::Ptr{Cvoid}, status::Ptr{Cvoid}, iters::Ptr{Cvoid},
                                       dz::Ptr{Cvoid}, du::Ptr{Cvoid}, stream::Ptr{Cvoid})::Cint)
end

"environment groups of step_dev! (0: the library's choice, 1: one launch on the caller's stream); needs GPU_MAX_HW_QUEUES >= groups + 1"
set_groups!(bm::BatchedMechanism, n::Integer) = check(@ccall $(fn(:dojo_set_groups))(bm.handle::Ptr{Cvoid}, n::Int32)::Cint)
"iteration cap of the step kernel: solves unfinished after `cap` Newton iterations go on in the continuation kernel (line-search trials side by side) in joined steps; 0 / < 0: off (the default); include/dojo_hip.h has the measurements"
set_iteration_cap!(bm::BatchedMechanism, cap::Integer) = check(@ccall $(fn(:dojo_set_iteration_cap))(bm.handle::Ptr{Cvoid}, cap::Int32)::Cint)
set_dispatch_order!(bm::BatchedMechanism, mode::Integer) = check(@ccall $(fn(:dojo_set_dispatch_order))(bm.handle::Ptr{Cvoid}, mode::Int32)::Cint)
"asynchronous environment groups: consecutive step_dev! calls chain per group; join!(bm; stream) orders `stream` behind everything in flight"
set_async!(bm::BatchedMechanism, on::Bool) = check(@ccall $(fn(:dojo_set_async))(bm.handle::Ptr{Cvoid}, on::Int32)::Cint)
set_async!(bm::BatchedMechanism, mode::Integer) = check(@ccall $(fn(:dojo_set_async))(bm.handle::Ptr{Cvoid}, Int32(mode)::Int32)::Cint)      # 2: pipelined groups (dojo_hip.h)
join!(bm::BatchedMechanism; stream::DevPtr=C_NULL) = check(@ccall $(fn(:dojo_join))(bm.handle::Ptr{Cvoid}, stream::Ptr{Cvoid})::Cint)

"""
    set_refinement!(bm, stiffness)

Iterative refinement of the linear solves (Newton directions and IFT columns) of every environment whose cones reach
max γ/s > `stiffness` (Inf: never, 0: always).  Default: 1e4 when `opts.rtol <= 1e-7` or `opts.btol <= 1e-6`, never at the
reference's default tolerances (DESIGN.md §4.5).
"""
set_refinement!(bm::BatchedMechanism, stiffness::Real) = check(@ccall $(fn(:dojo_set_refinement))(bm.handle::Ptr{Cvoid}, Float64(stiffness)::Cdouble)::Cint)

# Multi-GPU: one Julia process per GPU (Distributed / MPI.jl), each with ONE BatchedMechanism for its contiguous slice of
# the batch.  Rank 0 creates the 128-byte id, the host carries it to the other ranks, every rank joins; allgather! then
# moves per-rank device buffers over RCCL / xGMI in rank (= batch) order -- once per rollout chunk, not per step.
#   id = myrank == 0 ? DojoHIP.comm_unique_id() : nothing
#   id = MPI.bcast(id, 0, comm)                        # or Distributed.remotecall_fetch
#   DojoHIP.comm_init!(bm, myrank, nranks, id)
#   DojoHIP.allgather!(bm, pointer(z_local), pointer(z_all), length(z_local))
"128 opaque bytes identifying a new RCCL communicator (call on rank 0 only)"
function comm_unique_id()
    id = Vector{UInt8}(undef, 128)
    check(@ccall $(fn(:dojo_comm_unique_id))(id::Ptr{UInt8})::Cint)
    return id
end
comm_init!(bm::BatchedMechanism, rank::Integer, world::Integer, id::Vector{UInt8}) =
    check(@ccall $(fn(:dojo_comm_init))(bm.handle::Ptr{Cvoid}, rank::Int32, world::Int32, id::Ptr{UInt8})::Cint)
"recv[count, world] <- send[count] of every rank; elements of the handle's dtype, or Int32 with `as_int32` (status / iteration buffers)"
allgather!(bm::BatchedMechanism, send::DevPtr, recv::DevPtr, count::Integer; as_int32::Bool=false, stream::DevPtr=C_NULL) =
    check(@ccall $(fn(:dojo_allgather_dev))(bm.handle::Ptr{Cvoid}, send::Ptr{Cvoid}, recv::Ptr{Cvoid}, count::Int64, as_int32::Int32, stream::Ptr{Cvoid})::Cint)
"(rank, world) of the handle's communicator, (0, 1) before comm_init!"
function comm_info(bm::BatchedMechanism)
    r = Ref{Int32}(0); w = Ref{Int32}(1)
    check(@ccall $(fn(:dojo_comm_info))(bm.handle::Ptr{Cvoid}, r::Ref{Int32}, w::Ref{Int32})::Cint)
    return Int(r[]), Int(w[])
end

# ---- opt-in single-Mechanism drop-in ------------------------------------------------------------
# DojoHIP.enable!(mechanism) makes Dojo.mehrotra!(mechanism) (src/solver/mehrotra.jl:9) round-trip through the library
# (B = 1, fp64) and write the solution back -- body.state.vsol/ωsol, joint.impulses, contact.impulses(_dual), mechanism.μ,
# and mechanism.system re-assembled at the solution -- so that step!, simulate!, get_maximal_gradients! and the
# DojoEnvironments built on them (DojoEnvironments/src/environments.jl:77-84) keep working unchanged.
# Mechanisms that were not enabled keep the reference's own mehrotra!.
const HANDLES = IdDict{Dojo.Mechanism,BatchedMechanism{Float64}}()
const OVERRIDE_WORLD = Ref{UInt}(0)

"""
install the Dojo.mehrotra! method that dispatches enabled mechanisms to the library (done once, by the first enable!).
UNVERIFIED (no Julia where this was written): the method is `@eval`ed at run time, so call `enable!` at top level before the first
`step!` (a caller compiled earlier keeps dispatching to Dojo's own method until it returns to top level), and never while precompiling.
"""
function install_override!()
    ccall(:jl_generating_output, Cint, ()) == 1 && error("DojoHIP.enable! must not run during precompilation")
    OVERRIDE_WORLD[] != 0 && return
    OVERRIDE_WORLD[] = Base.get_world_counter()           # the world in which Dojo's own method is still the one that runs
    @eval function Dojo.mehrotra!(mechanism::Dojo.Mechanism{T}; opts=Dojo.SolverOptions{T}()) where T
        bm = get(HANDLES, mechanism, nothing)
        bm === nothing && return Base.invoke_in_world(OVERRIDE_WORLD[], Dojo.mehrotra!, mechanism; opts=opts)
        return hip_mehrotra!(mechanism, bm; opts=opts)
    end
    return
end

function enable!(m::Dojo.Mechanism)
    HANDLES[m] = BatchedMechanism(m, 1; T=Float64)
    install_override!()
    return m
end
disable!(m::Dojo.Mechanism) = (delete!(HANDLES, m); m)

"""
    hip_mehrotra!(mechanism, bm; opts)

What `mehrotra!` finds when `step!` calls it (src/simulation/step.jl:11-30): `set_maximal_state!` has set x2, q2, v15, ω15 and
`set_input!` / `input_impulse!` (src/mechanism/set.jl:40-53) have folded the controls into every body's `state.JF2`,
`state.Jτ2` and CLEARED the joints' inputs -- so the controls cross the boundary as body impulses (`dojo_step_impulses`),
next to `state.Fext`, `state.τext`.
"""
function hip_mehrotra!(m::Dojo.Mechanism, bm::BatchedMechanism{Float64}; opts=Dojo.SolverOptions{Float64}())
    set_options!(bm, opts)
    Nb = length(m.bodies)
    z = reshape(Dojo.get_maximal_state(m), :, 1)
    jf = Matrix{Float64}(undef, 6Nb, 1); fext = Array{Float64,3}(undef, 6, Nb, 1)
    for (i, b) in enumerate(m.bodies)
        jf[6i-5:6i-3, 1] = b.state.JF2; jf[6i-2:6i, 1] = b.state.Jτ2
        fext[1:3, i, 1] = b.state.Fext; fext[4:6, i, 1] = b.state.τext
    end
    set_external_force!(bm, fext)
    zn = similar(z); status = Vector{Int32}(undef, 1); iters = Vector{Int32}(undef, 1)
    check(@ccall $(fn(:dojo_step_impulses))(bm.handle::Ptr{Cvoid}, z::Ptr{Float64}, jf::Ptr{Float64}, zn::Ptr{Float64}, status::Ptr{Int32}, iters::Ptr{Int32})::Cint)
    # ---- write-back (what mehrotra! mutates, SURVEY.md §8b) ----
    per = any(c -> c.model isa Dojo.LinearContact, m.contacts) ? 12 : 8      # exported [s; γ] scalars per contact (dojo_hip.h, dojo_get_solution)
    vel = Vector{Float64}(undef, 6Nb); ji = Vector{Float64}(undef, max(1, sum(length.(m.joints)))); cs = Vector{Float64}(undef, max(1, per * length(m.contacts)))
    check(@ccall $(fn(:dojo_get_solution))(bm.handle::Ptr{Cvoid}, vel::Ptr{Float64}, ji::Ptr{Float64}, cs::Ptr{Float64})::Cint)
    for (i, b) in enumerate(m.bodies)
        b.state.vsol[2] = SVector{3}(vel[6i-5:6i-3]); b.state.ωsol[2] = SVector{3}(vel[6i-2:6i])
        b.state.vsol[1] = b.state.vsol[2]; b.state.ωsol[1] = b.state.ωsol[2]        # update! (src/solver/linear_system.jl:32-45)
    end
    off = 0
    for j in m.joints
        n = length(j); j.impulses[2] = SVector{n}(ji[off+1:off+n]); j.impulses[1] = j.impulses[2]; off += n
    end
    for (i, c) in enumerate(m.contacts)
        nh = length(c.impulses[2])            # N½: 4 for NonlinearContact, 1 for ImpactContact (the device exports [s(4); γ(4)] for both), 6 for LinearContact ([s(6); γ(6)])
        o = per * (i - 1); h = per ÷ 2
        c.impulses_dual[2] = SVector{nh}(cs[o+1:o+nh]); c.impulses[2] = SVector{nh}(cs[o+h+1:o+h+nh])
        c.impulses_dual[1] = c.impulses_dual[2]; c.impulses[1] = c.impulses[2]
    end
    mu = Vector{Float64}(undef, 1)
    check(@ccall $(fn(:dojo_get_mu))(bm.handle::Ptr{Cvoid}, mu::Ptr{Float64})::Cint)
    m.μ = mu[1]
    Dojo.set_entries!(m)                      # mechanism.system: the un-factored Jacobian and residual at the solution (mehrotra.jl:69)
    status[1] == 2 && error("Excessive angular velocity.")       # src/solver/line_search.jl:18-20
    return status[1] == 0 ? :success : :failed
end

end # module
