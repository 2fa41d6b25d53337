"""Minimal Kaldi table/matrix/wave IO for the TEST infrastructure (oracle + fixtures).
Formats: base/io-funcs-inl.h:291-318 (binary header \\0B), matrix/kaldi-matrix.cc:1382-1400 (FM/DM),
feat/wave-reader.h (RIFF PCM16).  Not used by kaldi_amd/."""
import struct, numpy as np

def write_wav(path, samples_i16, rate=16000):
    data = np.asarray(samples_i16, dtype='<i2').tobytes()
    with open(path, 'wb') as f:
        f.write(b'RIFF' + struct.pack('<I', 36 + len(data)) + b'WAVE')
        f.write(b'fmt ' + struct.pack('<IHHIIHH', 16, 1, 1, rate, rate * 2, 2, 16))
        f.write(b'data' + struct.pack('<I', len(data)) + data)

def read_wav(path):
    b = open(path, 'rb').read()
    assert b[:4] == b'RIFF' and b[8:12] == b'WAVE'
    pos = 12; rate = None
    while pos < len(b):
        cid = b[pos:pos+4]; sz = struct.unpack('<I', b[pos+4:pos+8])[0]
        if cid == b'fmt ':
            fmt, ch, rate = struct.unpack('<HHI', b[pos+8:pos+16]); assert fmt == 1 and ch == 1
        elif cid == b'data':
            n = min(sz, len(b) - pos - 8)
            return np.frombuffer(b[pos+8:pos+8+n - (n % 2)], dtype='<i2').copy(), rate
        pos += 8 + sz + (sz & 1)
    raise ValueError('no data chunk')

def _read_token(f):
    t = b''
    while True:
        c = f.read(1)
        if c in (b' ', b''): return t
        t += c

def read_matrix_binary(f):
    tok = _read_token(f)
    assert tok in (b'FM', b'DM', b'FV', b'DV'), tok
    dt = '<f4' if tok[0:1] == b'F' else '<f8'
    if tok[1:2] == b'M':
        assert f.read(1) == b'\x04'; r = struct.unpack('<i', f.read(4))[0]
        assert f.read(1) == b'\x04'; c = struct.unpack('<i', f.read(4))[0]
        return np.frombuffer(f.read(r * c * int(dt[-1])), dtype=dt).reshape(r, c).copy()
    assert f.read(1) == b'\x04'; n = struct.unpack('<i', f.read(4))[0]
    return np.frombuffer(f.read(n * int(dt[-1])), dtype=dt).copy()

def read_ark(path):
    """binary ark of float matrices -> dict key -> ndarray"""
    out = {}
    with open(path, 'rb') as f:
        while True:
            key = _read_token(f)
            if not key: break
            assert f.read(2) == b'\0B'
            out[key.decode()] = read_matrix_binary(f)
    return out

def write_ark(path, d):
    with open(path, 'wb') as f:
        for k, m in d.items():
            m = np.ascontiguousarray(m, dtype='<f4')
            f.write(k.encode() + b' \0BFM ' + b'\x04' + struct.pack('<i', m.shape[0]) + b'\x04' + struct.pack('<i', m.shape[1]))
            f.write(m.tobytes())

def read_htk(path):
    """HTK parameter file (feat/feature-functions / util ReadHtk): big-endian header + float32."""
    b = open(path, 'rb').read()
    n, period, size, kind = struct.unpack('>iihh', b[:12])
    return np.frombuffer(b[12:12 + n * size], dtype='>f4').reshape(n, size // 4).astype(np.float32)
