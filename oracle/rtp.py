"""TEST INFRASTRUCTURE -- CPU restatement of the RTP ingress stage.  Only tests/ may import it.

PARITY UNPINNED: the reference delegates parsing and re-ordering to the third-party C extension `rtpsynth`
(`from rtpsynth.RtpJBuf import RtpJBuf, RTPFrameType, RTPParseError`, RTP/InfernRTPIngest.py:6; pulled by
requirements.txt, unpinned, not vendored, not installed here) and holds no test or vector for it.  What is
restated is (1) the RTP fixed header of RFC 3550 section 5.1 and (2) the contract visible at the call site,
RTP/InfernRTPIngest.py:76-96: `udp_in(datagram)` returns the frames that became ready, in strictly consecutive
extended-sequence order (asserted at :91); a gap that is abandoned is one `ERS` frame carrying `lseq_start`,
`lseq_end` and `ts_diff`, which the caller replaces by `codec.silence(ts_diff)` (:84-86), so `ts_diff` is the
missing duration in timestamp units; malformed input raises `RTPParseError` and leaves the buffer as it was (:78-80).
"""
import struct


class ParseError(Exception):
    pass


def build_packet(seq, ts, payload, pt=0, ssrc=0x1234, marker=0, csrc=(), ext=None, pad=0):
    """An RTP datagram (RFC 3550 5.1).  ext = (profile_id, bytes multiple of 4); pad = padding bytes (>=1)."""
    b0 = 0x80 | (0x20 if pad else 0) | (0x10 if ext is not None else 0) | len(csrc)
    out = struct.pack('!BBHII', b0, (marker << 7) | pt, seq & 0xffff, ts & 0xffffffff, ssrc)
    for c in csrc:
        out += struct.pack('!I', c)
    if ext is not None:
        out += struct.pack('!HH', ext[0], len(ext[1]) // 4) + ext[1]
    out += payload
    if pad:
        out += b'\0' * (pad - 1) + bytes([pad])
    return out


def parse(data):
    if len(data) < 12:
        raise ParseError('short')
    b0, b1, seq, ts, ssrc = struct.unpack('!BBHII', data[:12])
    if b0 >> 6 != 2:
        raise ParseError('version')
    cc = b0 & 15
    off = 12 + 4 * cc
    if off > len(data):
        raise ParseError('csrc')
    if b0 & 0x10:
        if off + 4 > len(data):
            raise ParseError('ext')
        off += 4 + 4 * struct.unpack('!H', data[off + 2:off + 4])[0]
        if off > len(data):
            raise ParseError('ext')
    plen = len(data) - off
    if b0 & 0x20:
        pad = data[-1] if plen > 0 else 0
        if pad == 0 or pad > plen:
            raise ParseError('pad')
        plen -= pad
    return dict(version=2, padding=(b0 >> 5) & 1, extension=(b0 >> 4) & 1, cc=cc, marker=b1 >> 7, pt=b1 & 127, seq=seq,
                ts=ts, ssrc=ssrc, payload_off=off, payload_len=plen)


class JBuf:
    """Zero-delay re-ordering buffer: see the module docstring.  udp_in -> list of ('rtp', lseq, ts, payload) /
    ('ers', lseq_start, lseq_end, ts_diff)."""

    def __init__(self, depth, ts_per_byte=1, max_payload=1472):
        self.depth, self.tpb, self.max_payload = depth, ts_per_byte, max_payload
        self.ref = None
        self.last = None
        self.next_ts = 0
        self.held = {}
        self.counts = dict(received=0, released=0, late=0, duplicate=0, reordered=0, ers_events=0, ers_packets=0,
                           ers_bytes=0, parse_errors=0)

    def _emit(self, out, lseq, ts, payload):
        self.last = lseq
        self.next_ts = (ts + len(payload) * self.tpb) & 0xffffffff
        self.counts['released'] += 1
        out.append(('rtp', lseq, ts, payload))

    def _flush(self, out):
        while self.last + 1 in self.held:
            ts, payload = self.held.pop(self.last + 1)
            self._emit(out, self.last + 1, ts, payload)

    def udp_in(self, data):
        try:
            h = parse(data)
            if h['payload_len'] > self.max_payload:
                raise ParseError('long')
        except ParseError:
            self.counts['parse_errors'] += 1
            raise
        self.counts['received'] += 1
        if self.ref is None:
            lseq = h['seq']
            self.ref = lseq
        else:
            d = (h['seq'] - (self.ref & 0xffff)) & 0xffff
            if d >= 0x8000:
                d -= 0x10000
            lseq = self.ref + d
            self.ref = max(self.ref, lseq)
        payload = data[h['payload_off']:h['payload_off'] + h['payload_len']]
        out = []
        if self.last is not None and lseq <= self.last:
            self.counts['late'] += 1
        elif self.last is None or lseq == self.last + 1:
            self._emit(out, lseq, h['ts'], payload)
            self._flush(out)
        elif lseq in self.held:
            self.counts['duplicate'] += 1
        else:
            self.held[lseq] = (h['ts'], payload)
            self.counts['reordered'] += 1
            while len(self.held) > self.depth:
                head = min(self.held)
                ts_diff = (self.held[head][0] - self.next_ts) & 0xffffffff
                out.append(('ers', self.last + 1, head - 1, ts_diff))
                self.counts['ers_events'] += 1
                self.counts['ers_packets'] += head - 1 - self.last
                self.last = head - 1
                self.next_ts = self.held[head][0]
                self._flush(out)
        return out


def released_bytes(frames, tpb=1, fill=0xff, fifo_cap=None):
    """What InfernRTPIngest.py:82-96 hands to VADChannel.ingest for these frames, concatenated."""
    out = b''
    for f in frames:
        if f[0] == 'ers':
            n = f[3] // tpb
            out += bytes([fill]) * (min(n, fifo_cap) if fifo_cap else n)
        else:
            out += f[3]
    return out
