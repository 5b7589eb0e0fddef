"""The GRU regressor of the reference's `examples/pydynet/ts_prediction.py` (GRU(batch_first) -> last
hidden state -> Linear, MSE loss, Adam lr 0.01) and the 1-D descent of `autograd1d.py`, stated once
and parameterised by the package namespace (real reference in tools/gen_golden.py, pydynet_amd in the
tests).  Sequence length / width are reduced (the example is T=40, H=32) to keep the fixture small."""
import numpy as np

CFG = dict(T=12, H=16, batch=24, lr=0.01, steps=3)


def build(pdn, nn):
    class RNN(nn.Module):
        def __init__(self):
            super().__init__()
            self.rnn = nn.GRU(input_size=1, hidden_size=CFG["H"], num_layers=1, batch_first=True, dtype=np.float32)
            self.out = nn.Linear(CFG["H"], 1, dtype=np.float32)

        def forward(self, x, h_state):
            _, h_state = self.rnn(x, h_state)
            return self.out(h_state[:, self.rnn.num_layers - 1, :])

    return RNN


def make_inputs():
    t = np.arange(0, 100, .05)
    y = np.sin(np.pi * t) + 0.5 * np.cos(2 * np.pi * t)
    T = CFG["T"]
    idx = np.arange(CFG["batch"])[:, None] * 3 + np.arange(T)[None, :]
    X = y[idx][..., None].astype(np.float32)                    # (B, T, 1)
    Y = y[idx[:, -1] + 1][:, None].astype(np.float32)           # (B, 1)
    return X, Y


def run(pdn, nn, Adam, device=None, to_host=lambda a: a):
    net = build(pdn, nn)()
    if device is not None:
        net = net.to(device)
    opt = Adam(net.parameters(), lr=CFG["lr"])
    crit = nn.MSELoss()
    X, Y = make_inputs()
    kw = {} if device is None else {"device": device}
    out, losses = {}, []
    for s in range(CFG["steps"]):
        pred = net(pdn.Tensor(X, dtype=np.float32, **kw), None)
        loss = crit(pred, pdn.Tensor(Y, dtype=np.float32, **kw))
        opt.zero_grad(); loss.backward(); opt.step()
        losses.append(loss.item())
        if s == 0:
            out["pred1"] = to_host(pred.data).copy()
            for name, p in net._parameters.items():
                if p.requires_grad:
                    out[f"grad1/{name}"] = to_host(p.grad).copy()
    out["losses"] = np.array(losses)
    return out


def autograd1d(pdn, device=None, x0=1.0, lr=1.5, n_iter=20):
    """autograd1d.py: descent on log((x - 7)^2 + 6) from x0 with the gradient from the tape."""
    kw = {} if device is None else {"device": device}
    x = pdn.Tensor(float(x0), requires_grad=True, **kw)
    xs = [float(x0)]
    for _ in range(n_iter):
        x.zero_grad()
        y = pdn.log((x - 7) ** 2 + 6)
        y.backward()
        with x.device:
            x.data -= lr * x.grad
        xs.append(x.item())
    return np.array(xs)
