"""Pin the recurrent-PPO restatements of the CPU oracle (GRU cell, RND reward, L4 loss) to
vectors captured from the reference's own classes (tests/golden/ppo_lstm_parts.npz)."""
import numpy as np

from conftest import load_golden, rel_close


def _gru_params(g):
    return (g["gru_sd_rnn.weight_ih_l0"], g["gru_sd_rnn.weight_hh_l0"], g["gru_sd_rnn.bias_ih_l0"], g["gru_sd_rnn.bias_hh_l0"])


def test_gru_matches_reference_urnn(oracle):
    """Forward of the reference URNN (torch.nn.GRU, batch_first) and its gradients by backpropagation through time
    with orc_gru_cell_bwd + float64 GEMMs."""
    g = load_golden("ppo_lstm_parts")
    w_ih, w_hh, b_ih, b_hh = _gru_params(g)
    x, h0 = g["gru_x"], g["gru_h0"]
    out, hn = oracle.gru_forward(x, h0, w_ih, w_hh, b_ih, b_hh)
    assert rel_close(out, g["gru_out"], 2e-6) <= 2e-6 and rel_close(hn, g["gru_hn"], 2e-6) <= 2e-6
    B, L, _ = x.shape
    f64 = lambda a: a.astype(np.float64)                      # noqa: E731
    hs, gis, ghs, h = [h0], [], [], h0
    for step in range(L):
        gis.append((f64(x[:, step]) @ f64(w_ih).T + b_ih).astype(np.float32))
        ghs.append((f64(h) @ f64(w_hh).T + b_hh).astype(np.float32))
        h = oracle.gru_cell_fwd(gis[-1], ghs[-1], h)
        hs.append(h)
    dh = f64(g["gru_w_h"])
    dW_ih, dW_hh = np.zeros_like(f64(w_ih)), np.zeros_like(f64(w_hh))
    db_ih, db_hh = np.zeros_like(f64(b_ih)), np.zeros_like(f64(b_hh))
    dx = np.zeros_like(f64(x))
    for step in reversed(range(L)):
        dh_out = (dh + f64(g["gru_w_out"][:, step])).astype(np.float32)
        dgi, dgh, dhd = oracle.gru_cell_bwd(gis[step], ghs[step], hs[step], dh_out)
        dW_ih += f64(dgi).T @ f64(x[:, step]); db_ih += f64(dgi).sum(0)
        dW_hh += f64(dgh).T @ f64(hs[step]); db_hh += f64(dgh).sum(0)
        dx[:, step] = f64(dgi) @ f64(w_ih)
        dh = f64(dhd) + f64(dgh) @ f64(w_hh)
    for got, key in ((dx, "gru_dx"), (dh, "gru_dh0"), (dW_ih, "gru_grad_rnn.weight_ih_l0"), (dW_hh, "gru_grad_rnn.weight_hh_l0"),
                     (db_ih, "gru_grad_rnn.bias_ih_l0"), (db_hh, "gru_grad_rnn.bias_hh_l0")):
        assert rel_close(got, g[key], 1e-5) <= 1e-5, key


def test_rnd_reward_matches_numpy_mean(oracle):
    g = load_golden("ppo_lstm_parts")
    rnd, rew = oracle.rnd_reward(g["rnd_predict"], g["rnd_target"], rew=np.arange(7, dtype=np.float32))
    assert rel_close(rnd, g["rnd_reward"], 1e-6) <= 1e-6
    assert np.array_equal(rew, np.arange(7, dtype=np.float32) + rnd)


def test_rnn_loss_matches_reference(oracle):
    """L4: gradients and masked-mean metrics of ppo_lstm_lunarlander.py:716-776; case 2 has an empty mask
    (every masked mean is 0 and no gradient flows)."""
    g = load_golden("ppo_lstm_parts")
    cfg = tuple(float(x) for x in g["cfg"])
    for c in range(int(g["n_cases"])):
        p = f"l{c}_"
        dl, dv, met = oracle.ppo_rnn_loss_fwd_bwd(g[p + "logits"], g[p + "values"], g[p + "actions"], g[p + "old_lp"],
                                                  g[p + "old_ent"], g[p + "old_values"], g[p + "adv"], g[p + "ret"], cfg)
        want = g[p + "metrics"]
        B = len(g[p + "values"])
        assert met[9] == want[7], c
        assert np.allclose(dl, g[p + "dlogits"], rtol=2e-5, atol=2e-7), c
        assert np.allclose(dv, g[p + "dvalues"], rtol=2e-5, atol=2e-7), c
        cnt = max(met[9], 1.0)
        got = np.array([met[0] / cnt, met[1] / cnt, met[2] / cnt, met[3] / cnt, met[4] / B, met[5] / B,
                        (met[8] - met[6] * met[7] / B) / B])
        assert rel_close(got, want[:7], 1e-5) <= 1e-5, (c, got, want)
    assert float(g["l2_metrics"][7]) == 0.0 and not g["l2_dlogits"].any()


def test_ppo_full_pscn_network_matches_reference():
    """ppo_full's ActorCritic with use_mhc = False (PSCN trunk): same parameter names, same initial weights from the
    same torch seed (construction order), bit-identical forward on CPU."""
    import torch
    from gymrl_amd.ppo_full_lunarlander import ActorCritic, Config
    g = load_golden("ppo_full_pscn")
    cfg = Config()
    cfg.use_mhc = False
    torch.manual_seed(77)
    net = ActorCritic(8, 4, cfg)
    assert list(net.state_dict().keys()) == [str(k) for k in g["keys"]]
    chk = np.array([float(v.double().sum()) for v in net.state_dict().values()])
    assert np.array_equal(chk, g["checksum"])
    logits, values = net(torch.from_numpy(g["x"]))
    assert np.array_equal(logits.detach().numpy(), g["logits"]) and np.array_equal(values.detach().numpy(), g["values"])
