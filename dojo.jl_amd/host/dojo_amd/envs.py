"""DojoEnvironments' `Environment` API (DojoEnvironments/src/environments.jl:60-110) for B copies of an environment on
one GPU: `step`, `get_state`, `state_map`, `input_map` of `AntARS` (environments/ant_ars.jl) and `QuadrupedSampling`
(environments/quadruped_sampling.jl).  States, inputs and observations are torch tensors on the device; every call
enqueues kernels of libdojo_hip.so on torch's current stream and returns without synchronizing, so a sampling loop
(observation -> policy -> step -> reward) never touches the host."""
import ctypes as C
import numpy as np
from . import api
from .coords import nominal_minimal
from .mechanisms import get_mechanism
from .topology import SolverOptions

# environment -> (mechanism, contact forces in the observation, unactuated leading inputs)
_ENVIRONMENTS = {"ant_ars": ("ant", True, 6), "quadruped_sampling": ("quadruped", False, 6)}


class BatchedEnvironment:
    def __init__(self, name, batch, dtype="f32", device=0, opts=None, **mechanism_kwargs):
        import torch
        mech, self.contact_forces, self.n_unactuated = _ENVIRONMENTS[name]
        self.name = name
        self.spec = get_mechanism(mech, **mechanism_kwargs)
        self.batch = int(batch)
        self.device = torch.device("cuda", device)
        self.mechanism = api.BatchedMechanism(self.spec, batch, dtype=dtype, device=device, opts=opts or SolverOptions())
        self.torch_dtype = torch.float32 if self.mechanism.np_dtype == np.float32 else torch.float64
        self.nx = 2 * self.spec.nu
        self.nobs = self.nx + (len(self.spec.contacts) if self.contact_forces else 0)
        self.status = torch.zeros(self.batch, dtype=torch.int32, device=self.device)
        self.iters = torch.zeros(self.batch, dtype=torch.int32, device=self.device)
        self._x = torch.zeros(self.batch, self.nx, dtype=self.torch_dtype, device=self.device)      # minimal state after the last step
        self._stepped = False

    # ---- ant_ars.jl:53-61, quadruped_sampling.jl:51-58 ----
    def state_map(self, state):
        return state[:, :self.nx]

    def input_map(self, inp):
        import torch
        return torch.cat([torch.zeros(self.batch, self.n_unactuated, dtype=self.torch_dtype, device=self.device), inp.to(self.torch_dtype)], dim=1)

    def _stream(self):
        import torch
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def initialize(self, x0=None, **kw):
        """initialize!(environment, model; kwargs...) (environments.jl:112-114): every copy at the mechanism's nominal state
        (keywords as the reference's initialize_<model>!), or at the given minimal states x0 [B, 2nu]."""
        import torch
        if x0 is None:
            x0 = np.tile(nominal_minimal(self.spec, **kw), (self.batch, 1))
        self._x = torch.as_tensor(x0, dtype=self.torch_dtype).to(self.device).reshape(self.batch, self.nx).contiguous()
        self._stepped = False
        return self._x

    def step(self, state, inp):
        """step!(environment, state, input) (environments.jl:77-84): step_minimal_coordinates! of state_map(state), input_map(input)."""
        import torch
        x = self.state_map(state).to(self.torch_dtype).contiguous()
        u = self.input_map(inp).contiguous()
        xn = torch.empty_like(x)
        api._chk(api.lib().dojo_step_minimal_dev(self.mechanism.h, C.c_void_p(x.data_ptr()), C.c_void_p(u.data_ptr()), C.c_void_p(xn.data_ptr()),
                                                 C.c_void_p(self.status.data_ptr()), C.c_void_p(self.iters.data_ptr()), self._stream()))
        self._x = xn
        self._stepped = True
        self._keep = (x, u)                                     # inputs of kernels still in flight
        return xn

    def get_state(self):
        """get_state(environment): minimal state (+ clamped contact normal impulses for AntARS, ant_ars.jl:72-80)."""
        import torch
        if not self._stepped:        # a freshly built ContactConstraint holds the neutral vector [1,1,0,0] (contacts/constructor.jl:38-39)
            obs = torch.ones(self.batch, self.nobs, dtype=self.torch_dtype, device=self.device)
            obs[:, :self.nx] = self._x
            return obs
        obs = torch.empty(self.batch, self.nobs, dtype=self.torch_dtype, device=self.device)
        api._chk(api.lib().dojo_observe_dev(self.mechanism.h, None, C.c_void_p(obs.data_ptr()), int(self.contact_forces), self._stream()))
        return obs

    def close(self):
        self.mechanism.close()
