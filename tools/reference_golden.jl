# reference_golden.jl -- run the REFERENCE (Dojo.jl) on this repository's seeded inputs and write what it computes.
#
# The build container has no Julia (SURVEY.md §8c), so every parity number in this repository is "HIP path vs the C++ oracle".
# This script is the other half: on any machine with Julia >= 1.6 and a checkout of dojo-sim/Dojo.jl,
#
#     python tools/reference_exchange.py export                           # (here) tests/golden/reference_inputs/config<N>.txt
#     julia --project=/path/to/Dojo.jl tools/reference_golden.jl          # (there; DojoEnvironments dev'ed into the project)
#     python -m pytest tests/test_reference_golden.py                     # (here) oracle and HIP path against the reference
#
# writes tests/golden/reference_outputs/config<N>.txt.  Everything travels by BODY / JOINT NAME: the reference orders its
# bodies by Julia Dict iteration (src/mechanism/urdf.jl), which is not reproducible outside Julia.
# Only the standard library is used besides Dojo / DojoEnvironments.
#
# What is run, per case (the seams of SURVEY.md §8b):
#     z_next = step!(mechanism, z, u; opts)                       src/simulation/step.jl:11-30
#     zn     = get_maximal_state(mechanism)                       the mechanism's state after update_state! (what dojo_step returns;
#                                                                 step!'s own return value is that state advanced once more, Q1)
#     jz, ju = get_maximal_gradients!(mechanism, z, u; opts)      src/gradients/state.jl:69-126 (literal: DOJO_GRAD_REFERENCE)
using Dojo
using DojoEnvironments
using Printf

const ROOT = normpath(joinpath(@__DIR__, ".."))
const INDIR = joinpath(ROOT, "tests", "golden", "reference_inputs")
const OUTDIR = joinpath(ROOT, "tests", "golden", "reference_outputs")

parse_kw(s) = begin
    kw = Dict{Symbol,Any}()
    for item in split(s)
        k, v = split(item, "=")
        kw[Symbol(k)] = v == "true" ? true : v == "false" ? false : tryparse(Float64, v) === nothing ? String(v) : (occursin(".", v) ? parse(Float64, v) : parse(Int, v))
    end
    kw
end

# get_two_body of test/collisions.jl:2-58 with named bodies: a sphere on a joint to the world and a free second sphere that touches it
# through a SphereSphereCollision contact (forward only: the reference has no data Jacobians for it)
function two_spheres(; friction_type="nonlinear", joint_world_body1="Floating", gravity=-9.81, timestep=0.1, radius=0.5, mass=1.0, friction_coefficient=0.5)
    origin = Dojo.Origin{Float64}()
    pbody = Dojo.Sphere(radius, mass; name=:sphere1)
    cbody = Dojo.Sphere(radius, mass; name=:sphere2)
    joint = Dojo.JointConstraint((joint_world_body1 == "Fixed" ? Dojo.Fixed : Dojo.Floating)(origin, pbody); name=:joint)
    if friction_type == "impact"
        collision = Dojo.SphereSphereCollision{Float64,0,3,0}(zeros(3), zeros(3), radius, radius)
        model = Dojo.ImpactContact{Float64,2}(zeros(Float64, 0, 2), collision)
    elseif friction_type == "linear"
        collision = Dojo.SphereSphereCollision{Float64,2,3,6}(zeros(3), zeros(3), radius, radius)
        model = Dojo.LinearContact{Float64,12}(friction_coefficient, [0.0 1.0; 0.0 -1.0; 1.0 0.0; -1.0 0.0], collision)
    else
        collision = Dojo.SphereSphereCollision{Float64,2,3,6}(zeros(3), zeros(3), radius, radius)
        model = Dojo.NonlinearContact{Float64,8}(friction_coefficient, [1.0 0.0; 0.0 1.0], collision)
    end
    contacts = [Dojo.ContactConstraint((model, pbody.id, cbody.id), name=:body_body)]
    return Dojo.Mechanism(origin, [pbody, cbody], [joint], contacts; gravity=gravity, timestep=timestep)
end

function build(name::AbstractString, kw::Dict{Symbol,Any})
    name == "two_spheres" && return two_spheres(; Dict(k => (v isa Number ? v : String(v)) for (k, v) in kw)...)
    corners = pop!(kw, :contact_corners, nothing)           # not a keyword of the reference's get_block: it always builds 8 corners
    mech = DojoEnvironments.get_mechanism(Symbol(name); kw...)
    if corners !== nothing
        # BASELINE configs[1]: the four bottom corners only (contact1 .. contact4 of DojoEnvironments/src/mechanisms/block/mechanism.jl:43-59)
        mech = Mechanism(mech.origin, mech.bodies, mech.joints, mech.contacts[1:corners];
            gravity=mech.gravity, timestep=mech.timestep, input_scaling=mech.input_scaling)
    end
    return mech
end

fmt(v) = join((@sprintf("%.17g", x) for x in v), " ")

function run_config(path::AbstractString)
    lines = [split(l) for l in eachline(path) if !isempty(strip(l))]
    cfg = parse(Int, lines[1][2]); name = String(lines[1][3])
    kw = parse_kw(join(lines[1][4:end], " "))
    rtol, btol = parse(Float64, lines[2][2]), parse(Float64, lines[2][3])
    mech = build(name, kw)
    opts = SolverOptions(rtol=rtol, btol=btol)
    Nb = length(mech.bodies)
    bodyidx = Dict(String(b.name) => i for (i, b) in enumerate(mech.bodies))
    # inputs of a joint sit at the joint's slice of u in mechanism.joints order (src/mechanism/set.jl:33-53)
    joff = Dict{String,UnitRange{Int}}(); off = 0
    for j in mech.joints
        n = Dojo.input_dimension(j); joff[String(j.name)] = off+1:off+n; off += n
    end
    nu = off
    single_joint = length(mech.joints) == 1                   # (pendulum / block: the only joint, whatever it is called)
    cases = sort(unique(parse(Int, l[2]) for l in lines if l[1] == "case"))
    mkpath(OUTDIR)
    open(joinpath(OUTDIR, "config$(cfg).txt"), "w") do io
        println(io, "config $cfg $name ", join(lines[1][4:end], " "))
        for c in cases
            z = zeros(13Nb); u = zeros(nu)
            for l in lines
                (l[1] == "z" && parse(Int, l[2]) == c) || continue
                i = bodyidx[String(l[3])]
                z[13(i-1)+1:13i] = parse.(Float64, l[4:16])
            end
            for l in lines
                (l[1] == "u" && parse(Int, l[2]) == c) || continue
                r = single_joint ? (1:nu) : joff[String(l[3])]
                u[r] = parse.(Float64, l[4:3+length(r)])
            end
            forward_only = name == "two_spheres"
            jz, ju = forward_only ? (zeros(12Nb, 12Nb), zeros(12Nb, max(nu, 1))) : get_maximal_gradients!(mech, z, u; opts=opts)       # step! + IFT (the step is repeated below for the status)
            set_maximal_state!(mech, z); set_input!(mech, u)
            status = Dojo.mehrotra!(mech; opts=opts)
            for body in mech.bodies
                Dojo.update_state!(body, mech.timestep)
            end
            zn = get_maximal_state(mech)
            println(io, "status $c ", status == :success ? "success" : "failed")
            for (i, b) in enumerate(mech.bodies)
                println(io, "zn $c $(b.name) ", fmt(zn[13(i-1)+1:13i]))
            end
            for (i, bi) in enumerate(mech.bodies), (k, bk) in enumerate(mech.bodies)
                blk = jz[12(i-1)+1:12i, 12(k-1)+1:12k]
                any(!iszero, blk) && println(io, "dz $c $(bi.name) $(bk.name) ", fmt(permutedims(blk)))   # row-major
            end
            for (i, bi) in enumerate(mech.bodies), j in mech.joints
                r = joff[String(j.name)]; isempty(r) && continue
                blk = ju[12(i-1)+1:12i, r]
                jn = single_joint ? first(lines[k][3] for k in eachindex(lines) if lines[k][1] == "u") : j.name
                any(!iszero, blk) && println(io, "du $c $(bi.name) $(jn) ", fmt(permutedims(blk)))
            end
        end
    end
    println("wrote ", joinpath(OUTDIR, "config$(cfg).txt"))
end

for f in sort(readdir(INDIR))
    endswith(f, ".txt") && run_config(joinpath(INDIR, f))
end
