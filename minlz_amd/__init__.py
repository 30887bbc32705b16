"""minlz_amd — MI355X-native MinLZ block codec (host-side mirror of the reference's Go API).

The names follow the reference package (minio/minlz): Encode / Decode / AppendEncoded /
TryEncode / MaxEncodedLen / DecodedLen / IsMinLZ (encode.go:74-244, decode.go:50-156) and the
error values ErrCorrupt / ErrTooLarge / ErrUnsupported / ErrInvalidLevel / ErrCRC
(decode.go:29-40), raised as exceptions.  All compute runs in the HIP library behind the C ABI
of include/minlz_hip.h; nothing here falls back to a CPU codec.
"""
from .api import *  # noqa: F401,F403
