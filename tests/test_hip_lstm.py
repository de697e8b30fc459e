"""HIP LSTM encoder kernel (fp32 MFMA, through the C ABI) vs the CPU oracle's explicit-gate LSTM.  GPU only.

Tolerance: |h - h_oracle| <= 2e-5 (fp32 accumulation order differs: MFMA k-ordered fma chain vs BLAS blocking;
gates use <= 2-ulp hardware exp/rcp)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle.encoder import EncoderLSTMOracle


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _inputs(T, B, obs, seed, ad=1):
    gen = torch.Generator().manual_seed(seed)
    x = torch.randn(T, B, obs, generator=gen)
    a = torch.rand(T, B, ad, generator=gen) * (torch.rand(T, B, ad, generator=gen) < 0.1).float() if ad else None
    m = (torch.rand(T, B, obs, generator=gen) < 0.5).float()
    return x, a, m


@pytest.mark.parametrize("obs,H,B,T", [(80, 160, 100, 12), (40, 80, 77, 9), (80, 160, 1000, 5), (24, 44, 50, 7), (80, 160, 16, 3),
                                       (80, 160, 33, 1), (37, 75, 53, 6), (10, 157, 40, 4)])  # last two: obs % 4 != 0, H % 4 != 0
def test_final_state_matches_oracle(obs, H, B, T):
    from hode.lstm import lstm_final_state
    dev = _dev()
    torch.manual_seed(obs + B)
    enc = EncoderLSTMOracle(obs + 1, H, 12)
    with torch.no_grad():
        for p in enc.lstm.parameters():
            p.mul_(1.5)
    x, a, m = _inputs(T, B, obs, seed=B)
    with torch.no_grad():
        h_o, c_o = enc.final_hidden(x, a, m)
    p = enc.lstm
    h, c = lstm_final_state(x.to(dev), a.to(dev), m.to(dev), p.weight_ih_l0.to(dev), p.weight_hh_l0.to(dev),
                            p.bias_ih_l0.to(dev), p.bias_hh_l0.to(dev), reverse=True)
    assert (h.cpu() - h_o).abs().max().item() <= 2e-5, (h.cpu() - h_o).abs().max().item()
    assert (c.cpu() - c_o).abs().max().item() <= 5e-5


@pytest.mark.parametrize("nt", [1, 2, 3, 4])
def test_every_patient_tile_variant(nt, monkeypatch):
    """The library picks the patient tile (16*NT) from the batch size; force each compiled variant on a ragged batch."""
    from hode.lstm import lstm_final_state
    dev = _dev()
    monkeypatch.setenv("HODE_LSTM_NT", str(nt))
    obs, H, B, T = 80, 160, 16 * nt * 3 + 5, 4
    torch.manual_seed(nt)
    enc = EncoderLSTMOracle(obs + 1, H, 12)
    x, a, m = _inputs(T, B, obs, seed=nt)
    with torch.no_grad():
        h_o, c_o = enc.final_hidden(x, a, m)
    p = enc.lstm
    h, c = lstm_final_state(x.to(dev), a.to(dev), m.to(dev), p.weight_ih_l0.to(dev), p.weight_hh_l0.to(dev),
                            p.bias_ih_l0.to(dev), p.bias_hh_l0.to(dev), reverse=True)
    assert (h.cpu() - h_o).abs().max().item() <= 2e-5 and (c.cpu() - c_o).abs().max().item() <= 5e-5


def test_forward_time_unmasked_variant():
    """EncoderLSTMReal semantics: forward time order, no input masking, all inputs in one tensor."""
    from hode.lstm import lstm_final_state
    from oracle.encoder import lstm_cell
    dev = _dev()
    T, B, I, H = 6, 40, 37, 44
    torch.manual_seed(5)
    lstm = torch.nn.LSTM(I, H)
    x = torch.randn(T, B, I)
    with torch.no_grad():
        out, (h_o, c_o) = lstm(x)
    h, c = lstm_final_state(x.to(dev), None, None, lstm.weight_ih_l0.to(dev), lstm.weight_hh_l0.to(dev),
                            lstm.bias_ih_l0.to(dev), lstm.bias_hh_l0.to(dev), reverse=False)
    assert (h.cpu() - h_o[0]).abs().max().item() <= 2e-5


@pytest.mark.parametrize("obs,H,B,T", [(80, 160, 100, 8), (40, 80, 77, 6), (24, 44, 21, 5), (80, 160, 500, 3), (37, 75, 53, 4),
                                       (10, 157, 40, 3)])  # last two: the scalar staging / store paths (obs, H not multiples of 4)
def test_backward_matches_oracle_autograd(obs, H, B, T):
    """BPTT kernel + GEMMs vs autograd through the oracle's explicit-gate LSTM: rel-L2 <= 1e-4 on every parameter."""
    from hode.lstm import lstm_encode
    dev = _dev()
    torch.manual_seed(obs + B + 1)
    enc = EncoderLSTMOracle(obs + 1, H, 12)
    x, a, m = _inputs(T, B, obs, seed=B + 1)
    cot = torch.randn(B, H, generator=torch.Generator().manual_seed(2))
    h_o, _ = enc.final_hidden(x, a, m)
    (h_o * cot).sum().backward()
    p = enc.lstm
    prm = [q.detach().clone().to(dev).requires_grad_(True) for q in (p.weight_ih_l0, p.weight_hh_l0, p.bias_ih_l0, p.bias_hh_l0)]
    h = lstm_encode(x.to(dev), a.to(dev), m.to(dev), *prm, reverse=True)
    assert (h.detach().cpu() - h_o.detach()).abs().max().item() <= 2e-5
    (h * cot.to(dev)).sum().backward()
    for q, ref, name in zip(prm, (p.weight_ih_l0, p.weight_hh_l0, p.bias_ih_l0, p.bias_hh_l0), ("w_ih", "w_hh", "b_ih", "b_hh")):
        num = (q.grad.cpu().double() - ref.grad.double()).norm()
        den = ref.grad.double().norm()
        assert float(num / den) <= 1e-4, (name, float(num / den))


@pytest.mark.parametrize("nt", [1, 2, 3])
def test_backward_every_tile_variant(nt, monkeypatch):
    from hode.lstm import lstm_encode
    dev = _dev()
    monkeypatch.setenv("HODE_LSTM_NT", str(nt))
    obs, H, B, T = 80, 160, 16 * nt * 2 + 3, 4
    torch.manual_seed(nt + 10)
    enc = EncoderLSTMOracle(obs + 1, H, 12)
    x, a, m = _inputs(T, B, obs, seed=nt + 10)
    cot = torch.randn(B, H, generator=torch.Generator().manual_seed(3))
    h_o, _ = enc.final_hidden(x, a, m)
    (h_o * cot).sum().backward()
    p = enc.lstm
    prm = [q.detach().clone().to(dev).requires_grad_(True) for q in (p.weight_ih_l0, p.weight_hh_l0, p.bias_ih_l0, p.bias_hh_l0)]
    h = lstm_encode(x.to(dev), a.to(dev), m.to(dev), *prm, reverse=True)
    (h * cot.to(dev)).sum().backward()
    for q, ref in zip(prm, (p.weight_ih_l0, p.weight_hh_l0, p.bias_ih_l0, p.bias_hh_l0)):
        assert float((q.grad.cpu().double() - ref.grad.double()).norm() / ref.grad.double().norm()) <= 1e-4


@pytest.mark.parametrize("H", [5, 16, 24, 32, 40, 64, 70, 96, 100, 128, 144])
def test_every_compiled_hidden_size_forward_and_backward(H):
    """Padded hidden sizes 16, 32, ..., 160 (csrc/hode_lstm_tpw.hip; a hidden size in between runs on the next compiled one
    with zero-padded units): `EncoderLSTM(input_dim, hidden_dim, ...)` accepts any hidden_dim up to 160 (reference
    model.py:384-406 takes any).  Final state and every parameter gradient vs the oracle."""
    from hode.lstm import lstm_encode
    dev = _dev()
    obs, B, T = 12, 37, 5
    torch.manual_seed(H)
    enc = EncoderLSTMOracle(obs + 1, H, 8)
    x, a, m = _inputs(T, B, obs, seed=H)
    cot = torch.randn(B, H, generator=torch.Generator().manual_seed(4))
    h_o, _ = enc.final_hidden(x, a, m)
    (h_o * cot).sum().backward()
    p = enc.lstm
    prm = [q.detach().clone().to(dev).requires_grad_(True) for q in (p.weight_ih_l0, p.weight_hh_l0, p.bias_ih_l0, p.bias_hh_l0)]
    h = lstm_encode(x.to(dev), a.to(dev), m.to(dev), *prm, reverse=True)
    assert (h.detach().cpu() - h_o.detach()).abs().max().item() <= 2e-5
    (h * cot.to(dev)).sum().backward()
    for q, ref, name in zip(prm, (p.weight_ih_l0, p.weight_hh_l0, p.bias_ih_l0, p.bias_hh_l0), ("w_ih", "w_hh", "b_ih", "b_hh")):
        assert float((q.grad.cpu().double() - ref.grad.double()).norm() / ref.grad.double().norm()) <= 1e-4, (H, name)


def test_hidden_size_above_the_largest_kernel_is_a_configuration_error():
    import hode
    from hode.lstm import lstm_final_state
    dev = _dev()
    lstm = torch.nn.LSTM(9, 176)
    x, a, m = _inputs(3, 8, 8, seed=0)
    with pytest.raises(hode.HodeConfigError, match="exceeds the largest compiled kernel"):
        lstm_final_state(x.to(dev), a.to(dev), m.to(dev), lstm.weight_ih_l0.to(dev), lstm.weight_hh_l0.to(dev),
                         lstm.bias_ih_l0.to(dev), lstm.bias_hh_l0.to(dev))


@pytest.mark.parametrize("obs,ad,masked", [(80, 1, True), (24, 0, True), (37, 0, False), (80, 1, False), (6, 3, True)])
def test_fill_operand_writes_x_times_mask_into_the_first_columns(obs, ad, masked):
    """hode_lstm_fill_operand: h_prev[t][b][:obs] = x * mask (x without a mask), every other column untouched -- bit for bit
    (one multiplication per element); 16-byte path (obs % 4 == 0) and the scalar one."""
    from hode import _lib as L
    from hode import lstm as hl
    dev = _dev()
    T, B, H = 5, 67, 32
    g = torch.Generator().manual_seed(obs)
    x = torch.randn(T, B, obs, generator=g).to(dev)
    a = torch.randn(T, B, ad, generator=g).to(dev) if ad else None
    m = (torch.rand(T, B, obs, generator=g) < 0.5).float().to(dev) if masked else None
    w_ih, w_hh = torch.randn(4 * H, obs + ad).to(dev), torch.randn(4 * H, H).to(dev)
    b = torch.zeros(4 * H, device=dev)
    W = (obs + ad + H + 1 + 3) // 4 * 4
    hp = torch.full((T, B, W), -7.0, device=dev)
    d = hl._desc(x, a, m, w_ih, w_hh, b, b, True, True)
    dummy = torch.empty(1, device=dev)
    d.h_out, d.c_out, d.h_prev = dummy.data_ptr(), dummy.data_ptr(), hp.data_ptr()
    L.check(L.lib().hode_lstm_fill_operand(d, hl._stream()), "hode_lstm_fill_operand")
    torch.cuda.synchronize()
    assert torch.equal(hp[:, :, :obs], x * m if masked else x)
    assert bool((hp[:, :, obs:] == -7.0).all())
