"""oracle/am_torch.py -- TEST / BASELINE INFRASTRUCTURE ONLY: the acoustic restatement of oracle/am_ref.py in torch-CPU
float32, the arithmetic type of the reference's TFLite float graph, for the CPU baseline of bench.py (SURVEY.md 8d item 2:
"torch-CPU fp32 with torch.set_num_threads(4) to mirror SetNumThreads(4), tflitemodelstate.cc:200").  Batch 1, one
utterance at a time, like the reference interpreter (modelstate.h:16 BATCH_SIZE = 1).  "Restatement, not TFLite": parity
unpinned like am_ref.py; tests/test_oracle_am.py checks it against am_ref.

Layer order and semantics: training/coqui_stt_training/deepspeech_model.py:66-89 (dense + clipped ReLU), :144-168
(LSTMCell, forget_bias 0, gates i, j, f, o), :171-263, :357 (softmax)."""
import numpy as np

from . import am_ref


def to_torch(w):
    import torch
    return {k: torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)) for k, v in w.items()}


def am_forward(windows, W, relu_clip=am_ref.RELU_CLIP):
    """windows [T, 494] float32 (numpy) -> probs [T, C] float32 (numpy).  W = to_torch(weights)."""
    import torch
    with torch.no_grad():
        x = torch.from_numpy(np.ascontiguousarray(windows, dtype=np.float32))
        clip = lambda a: torch.clamp(a, 0.0, relu_clip)
        l1 = clip(x @ W["layer_1/weights"] + W["layer_1/bias"])
        l2 = clip(l1 @ W["layer_2/weights"] + W["layer_2/bias"])
        l3 = clip(l2 @ W["layer_3/weights"] + W["layer_3/bias"])
        H = W["layer_1/bias"].shape[0]
        K = W["lstm/kernel"]
        Kx, KhT = K[:H], K[H:].t().contiguous()     # [4H, H]: the recurrent product is a matrix-vector product per step
        xproj = l3 @ Kx + W["lstm/bias"]
        T = x.shape[0]
        c = torch.zeros(H); h = torch.zeros(H)
        hs = torch.empty(T, H)
        for t in range(T):
            z = torch.addmv(xproj[t], KhT, h)
            i, j, f, o = z[:H], z[H:2 * H], z[2 * H:3 * H], z[3 * H:]
            c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(j)
            h = torch.sigmoid(o) * torch.tanh(c)
            hs[t] = h
        l5 = clip(hs @ W["layer_5/weights"] + W["layer_5/bias"])
        logits = l5 @ W["layer_6/weights"] + W["layer_6/bias"]
        return torch.softmax(logits, dim=1).numpy()


def utterance_probs(audio_i16, W):
    feats = am_ref.MfccSpec().frames_fast(np.asarray(audio_i16, dtype=np.int16))
    return am_forward(am_ref.context_windows(feats), W)
