"""dojo_amd -- host-side mirror of Dojo.jl's Mechanism / step! / simulate! /
get_maximal_gradients! API in front of libdojo_hip.so (hand-written HIP for gfx950).

In production the host language is Julia (julia/DojoHIP.jl calls the same C ABI with
@ccall); Julia is not available in the build container, so this Python package is the
executable mirror the parity tests drive.
"""
from .topology import MechanismSpec, BodySpec, JointSpec, JointHalfSpec, ContactSpec, SolverOptions
from .mechanisms import (get_mechanism, get_pendulum, get_block, get_ant, get_quadruped, get_atlas, baseline_config,
                         get_slider, get_nslider, get_raiberthopper, get_npendulum, get_snake, get_twister, get_sphere, get_cartpole, get_block2d, get_dzhanibekov, get_tippetop, Prototype, set_limits, get_two_spheres, sphere_sphere_contact, add_limits, get_limited_chain, get_fourbar)
from .coords import (minimal_to_maximal, maximal_to_minimal, initialize, synthetic_inputs, nominal_minimal, fp32_abi_state)
