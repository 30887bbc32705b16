"""Stream framing restatement (writer.go sync path, reader.go Read): KATs + round trips. No GPU."""
import pytest

import oracle as O
from minlz_amd import synth

HDR4096 = bytes.fromhex("ff0600004d696e4c7a02")


def test_framing_kat():
    # TestReaderMinLZUncompressedDataOK, minlz_test.go:1120-1134
    s = HDR4096 + b"\x01\x08\x00\x00" + b"\x68\x10\xe6\xb6" + b"abcd" + b"\x20\x00\x00\x00"
    assert O.stream_decode(s, 100) == b"abcd"


def test_framing_errors():
    # minlz_test.go:1136-1182
    with pytest.raises(O.OracleError) as e:
        O.stream_decode(HDR4096 + b"\x01\x04\x00\x00", 100)
    assert e.value.code == O.ERR_CORRUPT
    hdr8m = bytes.fromhex("ff0600004d696e4c7a0d")
    n = (8 << 20) + 4
    body = bytes([1, n & 0xff, (n >> 8) & 0xff, (n >> 16) & 0xff]) + b"\x00" * n
    with pytest.raises(O.OracleError) as e:
        O.stream_decode(hdr8m + body, 9 << 20)
    assert e.value.code == O.ERR_CRC
    n += 1
    body = bytes([1, n & 0xff, (n >> 8) & 0xff, (n >> 16) & 0xff]) + b"\x00" * n
    with pytest.raises(O.OracleError) as e:
        O.stream_decode(hdr8m + body, 9 << 20)
    assert e.value.code == O.ERR_TOO_LARGE
    hdr1m = bytes.fromhex("ff0600004d696e4c7a0a")
    n = (1 << 20) + 1 + 4
    body = bytes([1, n & 0xff, (n >> 8) & 0xff, (n >> 16) & 0xff]) + b"\x00" * n
    with pytest.raises(O.OracleError) as e:
        O.stream_decode(hdr1m + body, 2 << 20)
    assert e.value.code == O.ERR_TOO_LARGE


def test_stream_header_bytes():
    s = O.stream_encode(b"x" * 5000, 1, 4096)
    assert s[:10] == HDR4096
    assert s[-6:] == b"\x20\x02\x00\x00\x88\x27"  # EOF chunk: 2-byte uvarint(5000)


@pytest.mark.parametrize("level", [0, 1, 2])
@pytest.mark.parametrize("bs", [4096, 65536, 1 << 20, 8 << 20])
def test_stream_roundtrip(level, bs):
    for d in (synth.text_like(300000, 2), synth.random_bytes(70000), synth.pattern("off2", 200001)):
        s = O.stream_encode(d, level, bs)
        assert O.stream_decode(s, d.size) == d.tobytes()


def test_stream_crc_detects_corruption():
    d = synth.text_like(100000, 4)
    s = bytearray(O.stream_encode(d, 1, 65536))
    s[40] ^= 1
    with pytest.raises(O.OracleError):
        O.stream_decode(bytes(s), d.size)
