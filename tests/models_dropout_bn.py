"""The three small classifiers of the reference's `examples/pydynet/dropout_bn.py` (plain, Dropout,
BatchNorm1d), stated once and parameterised by the package namespace so the SAME definition runs on
the real reference (tools/gen_golden.py) and on pydynet_amd (tests).  Widths are reduced (the
example is 4096-512-128-40) to keep the fixture small; the structure is the example's."""
import numpy as np

CFG = dict(d_in=256, h1=64, h2=32, classes=10, batch=40, lr=5e-5, steps=3, p_drop=0.05)


def build(pdn, nn, F):
    c = CFG

    class DNN(nn.Module):
        def __init__(self):
            super().__init__()
            self.fc1 = nn.Linear(c["d_in"], c["h1"], dtype=np.float32)
            self.fc2 = nn.Linear(c["h1"], c["h2"], dtype=np.float32)
            self.fc3 = nn.Linear(c["h2"], c["classes"], dtype=np.float32)

        def forward(self, x):
            return self.fc3(F.relu(self.fc2(F.relu(self.fc1(x)))))

    class DNNDropout(DNN):
        def __init__(self):
            super().__init__()
            self.dropout = nn.Dropout(p=c["p_drop"])

        def forward(self, x):
            x = F.relu(self.dropout(self.fc1(x)))
            x = F.relu(self.dropout(self.fc2(x)))
            return self.fc3(x)

    class DNNBN(DNN):
        def __init__(self):
            super().__init__()
            self.bn1 = nn.BatchNorm1d(c["h1"], dtype=np.float32)
            self.bn2 = nn.BatchNorm1d(c["h2"], dtype=np.float32)

        def forward(self, x):
            x = F.relu(self.bn1(self.fc1(x)))
            x = F.relu(self.bn2(self.fc2(x)))
            return self.fc3(x)

    return DNN, DNNDropout, DNNBN


def make_inputs():
    rng = np.random.default_rng(3)
    X = rng.random((CFG["batch"], CFG["d_in"])).astype(np.float32)
    y = rng.integers(0, CFG["classes"], CFG["batch"])
    return X, y


def run(pdn, nn, F, Adam, device=None, to_host=lambda a: a):
    """Three joint training steps exactly as the example does them ((l1 + l2 + l3).backward()), then an
    eval-mode forward; returns a dict of arrays."""
    nets = [cls() for cls in build(pdn, nn, F)]
    if device is not None:
        nets = [n.to(device) for n in nets]
    opts = [Adam(n.parameters(), lr=CFG["lr"]) for n in nets]
    loss_fn = nn.CrossEntropyLoss()
    X, y = make_inputs()
    kw = {} if device is None else {"device": device}
    out = {}
    losses = []
    for s in range(CFG["steps"]):
        for n in nets:
            n.train()
        np.random.seed(100 + s)                                  # the dropout masks come from the host RNG
        xb, yb = pdn.Tensor(X, dtype=np.float32, **kw), pdn.Tensor(y, dtype=np.int64, **kw)
        ls = [loss_fn(n(xb), yb) for n in nets]
        for o in opts:
            o.zero_grad()
        (ls[0] + ls[1] + ls[2]).backward()
        for o in opts:
            o.step()
        losses.append([l.item() for l in ls])
        if s == 0:
            for i, n in enumerate(nets):
                for name, p in n._parameters.items():
                    if p.requires_grad:
                        out[f"grad1/{i}/{name}"] = to_host(p.grad).copy()
    out["losses"] = np.array(losses)
    for n in nets:
        n.eval()
    with pdn.no_grad():
        xb = pdn.Tensor(X, dtype=np.float32, **kw)
        for i, n in enumerate(nets):
            out[f"eval/{i}"] = to_host(n(xb).data).copy()
    pdn.autograd.set_grad_enabled(True)
    for name in ("running_mean", "running_var"):
        out[f"bn1/{name}"] = to_host(getattr(nets[2].bn1, name).data).copy()
    return out
