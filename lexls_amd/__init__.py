"""lexls_amd — MI355X-native lexicographic-QR core behind the LexLSE/LexLSI interface of jrl-umi3218/lexls.

Only what the hot path needs: ``csrc/`` (hand-written HIP kernels for gfx950 + the C ABI of
include/lexls_hip.h), ``lexlse.BatchedLexLSE`` (host-side mirror of the reference's equality-solver
interface over that ABI) and ``problems`` (deterministic synthetic inputs, flop/byte model).
"""
from .lexlse import BatchedLexLSE  # noqa: F401
from .capi import LexlsError  # noqa: F401
