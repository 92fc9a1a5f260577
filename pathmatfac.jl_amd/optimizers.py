"""Optimizer descriptors.  AdaGrad mirrors Flux.Optimise.AdaGrad as built by construct_optimizer (src/fit.jl:41-43)
and applied through the SubArray overload of src/optimizers.jl:6-13 (one accumulator per parameter, initialised to
epsilon); Adam is the north-star's addition (Flux.Optimise.Adam semantics).  The state lives on the device; a new
optimizer object means fresh state (the reference re-creates the optimizer at every stage, src/fit.jl:55)."""


class AdaGrad:
    kind = "adagrad"

    def __init__(self, eta=0.1, epsilon=1e-8):
        self.eta = float(eta)
        self.epsilon = float(epsilon)
        self._bound_ctx = None   # id of the device context that holds this optimizer's accumulators

    def params(self):
        return dict(kind=self.kind, lr=self.eta, eps=self.epsilon)


class Adam:
    kind = "adam"

    def __init__(self, eta=0.001, beta=(0.9, 0.999), epsilon=1e-8):
        self.eta = float(eta)
        self.beta = (float(beta[0]), float(beta[1]))
        self.epsilon = float(epsilon)
        self._bound_ctx = None

    def params(self):
        return dict(kind=self.kind, lr=self.eta, eps=self.epsilon, beta1=self.beta[0], beta2=self.beta[1])
