"""ans_large_alphabet_amd — MI355X-native ANSfold / ANSrfold encode+decode path.

Product code: csrc/ (HIP kernels + C-ABI, built into libansx.so), include/ (C++17 mirror of the
reference's methods.hpp codec structs), and this ctypes host mirror.  Nothing here imports the
test oracle (oracle/), and there is no CPU fallback.
"""
from ._lib import (AnsxError, DEFAULT_BLOCK_INTS, DEFAULT_CKPT_INTERVAL, FLAG_COMPACT_ALPHABET, FOLD, INT, MSB,
                   NO_CHECKPOINTS, RFOLD, SINGLE_STREAM, build_library, lib)
from .codec import (ANSfold, ANSint, ANSmsb, ANSrfold, Context, generate_dev, generate_host, make_opts, parse_container, zipf_from_uniform,
                    parse_dist)

__all__ = ["ANSfold", "ANSrfold", "ANSmsb", "ANSint", "MSB", "INT", "FLAG_COMPACT_ALPHABET", "Context", "AnsxError", "build_library", "lib", "make_opts",
           "parse_container", "generate_dev", "generate_host", "zipf_from_uniform", "parse_dist", "FOLD", "RFOLD", "SINGLE_STREAM", "NO_CHECKPOINTS",
           "DEFAULT_BLOCK_INTS", "DEFAULT_CKPT_INTERVAL"]
