"""lnasr -- MI355X (gfx950) drop-in for the GMM-HMM hot path of yehuohan/ln-asr.

Same module / class / function surface as the reference package (`lnasr.hmm.HMM`,
`lnasr.gmmhmm.GMMHMM`, `lnasr.gmm.*`, `lnasr.mfcc.MFCC`, `lnasr.utils.*`,
`lnasr.recognizer.*`); the arithmetic runs in hand-written HIP kernels reached through a
ctypes C-ABI (`include/lnasr_hip.h`).  No PyTorch, no CPU fallback.
"""
__all__ = ["utils", "gmm", "hmm", "gmmhmm", "mfcc", "recognizer"]
