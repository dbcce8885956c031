"""Recognizer shell with the reference's constructor wiring (lnasr/recognizer.py:20-48).

The reference's `recognize()` is unimplemented (`pass`) and its language-model / lexicon side
is out of scope here.  `AcousticModel` wires MFCC + GMMHMM exactly as the
reference does; `Recognizer.recognize` runs the acoustic path the reference stops short of:
PCM -> MFCC -> Viterbi state path.  Upstream has nothing to compare with; the checker is the composition of the two
oracle-pinned halves, oracle.mfcc -> oracle.decode (tests/test_gpu_frontend.py::test_recognize_equals_oracle_chain).
The samples go to the device once; features, emissions and trellis stay in HBM and only the path [Nf] comes back.
"""
import numpy as np

from .mfcc import MFCC
from .gmmhmm import GMMHMM


class AcousticModel:
    def __init__(self, modelfile):
        self.mfcc = MFCC()
        self.gmmhmm = GMMHMM(n=2, m=2, d=self.mfcc.D)
        self.gmmhmm.load(modelfile)


class LanguageModel:
    """The n-gram language model is not part of the accelerated path; the model file is only recorded."""

    def __init__(self, modelfile):
        self.modelfile = modelfile
        self.ngram = None


class Recognizer:
    def __init__(self, model: dict):
        self.am = AcousticModel(model["am"])
        self.lm = LanguageModel(model["lm"]) if model.get("lm") is not None else None

    def recognize(self, data: np.ndarray):
        """PCM samples (int16 as read_pcm returns them, utils.py:100-104, or float64) -> most likely HMM state sequence
        (uint32 [Nf])."""
        data = np.ascontiguousarray(data)
        if data.dtype != np.int16:
            data = np.ascontiguousarray(data, dtype=np.float64)
        mf, gh = self.am.mfcc, self.am.gmmhmm
        ctx = gh.ctx
        cfg, win, bank = mf._cfg()
        d_pcm = ctx.to_device(data)
        _, _, feat, _ = ctx.mfcc(d_pcm, np.array([0, len(data)], dtype=np.int64), cfg, win, bank, False, False)
        return gh.decode(feat)
