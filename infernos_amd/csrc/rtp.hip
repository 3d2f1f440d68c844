// rtp.hip -- host-side RTP ingress stage in front of ifh_ingest_tick: RFC 3550 header
// parse, per-call re-ordering jitter buffer, loss fill, and the per-tick [n][frame_bytes] frame matrix.
//
// Replaces the per-packet Python of RTP/InfernRTPIngest.py:63-100 (one thread, one `RtpJBuf.udp_in` call
// and one `VADChannel.ingest` call per packet per call) by a table of calls advanced with plain C calls.
// The reference delegates the buffer itself to the third-party C extension `rtpsynth` (`RtpJBuf`, absent
// from the reference tree and from this image): the behaviour restated here is the one its call site relies
// on (InfernRTPIngest.py:76-96) -- frames come out in strictly consecutive extended sequence order, a gap
// that is given up on comes out as ONE erasure record {lseq_start, lseq_end, ts_diff} which the caller turns
// into `codec.silence(ts_diff)` bytes, a malformed datagram raises RTPParseError and changes nothing.
// PARITY UNPINNED against rtpsynth itself; cross-checked by tests/test_rtp_host.py.
//
// No device code in this file: it produces the host (pinned) frame matrix the tick kernel consumes.
#include <string.h>

#include <algorithm>
#include <new>
#include <vector>

#include "common.h"

namespace ifh {
namespace {

struct Held {           // a packet waiting for the gap in front of it to fill
    int64_t lseq;
    uint32_t ts;
    int32_t len;
    int32_t slab;       // index of its payload slab
    ifh_rtp_hdr hdr;
};

struct JStream {
    bool have_ref = false, have_out = false;
    int64_t ref_lseq = 0;       // highest extended sequence number seen (wrap reference)
    int64_t last_lseq = 0;      // last one released (valid when have_out)
    uint32_t next_ts = 0;       // timestamp the next in-order packet is expected to carry
    std::vector<Held> held;     // sorted by lseq, at most `depth` long between calls
    std::vector<int32_t> free_slabs;
    // released payload bytes, waiting to be cut into frames
    std::vector<uint8_t> fifo;
    int64_t head = 0, count = 0;
    int64_t stats[IFH_RTP_NSTATS] = {0};
};

struct RtpTable {
    int n = 0, depth = 0, frame_bytes = 0, ts_per_byte = 1, fill = 0xff;
    int64_t fifo_cap = 0;
    std::vector<JStream> st;
    std::vector<uint8_t> slabs;     // [n][depth + 1][IFH_RTP_MAX_PAYLOAD]
};

inline uint8_t *slab_ptr(RtpTable *t, int s, int k)
{
    return t->slabs.data() + ((size_t)s * (t->depth + 1) + k) * IFH_RTP_MAX_PAYLOAD;
}

void fifo_put(RtpTable *t, JStream &js, const uint8_t *src, int64_t len, int fill_byte)
{
    for (int64_t done = 0; done < len;) {
        if (js.count == t->fifo_cap) {           // consumer is not draining: the oldest byte goes
            js.head = (js.head + 1) % t->fifo_cap;
            js.count--;
            js.stats[IFH_RTP_STAT_OVERFLOW_BYTES]++;
        }
        const int64_t tail = (js.head + js.count) % t->fifo_cap;
        const int64_t run = std::min<int64_t>(std::min<int64_t>(len - done, t->fifo_cap - tail), t->fifo_cap - js.count);
        if (src) memcpy(js.fifo.data() + tail, src + done, (size_t)run);
        else memset(js.fifo.data() + tail, fill_byte, (size_t)run);
        js.count += run;
        done += run;
    }
}

struct Sink {           // where one push reports what it released
    ifh_rtp_rec *recs;
    int cap, n;
    uint8_t *payload;
    int64_t pcap, pn;
    bool truncated;
};

void emit_rtp(RtpTable *t, JStream &js, int stream, const Held &h, const uint8_t *data, Sink *out)
{
    js.have_out = true;
    js.last_lseq = h.lseq;
    js.next_ts = h.ts + (uint32_t)(h.len * t->ts_per_byte);
    js.stats[IFH_RTP_STAT_RELEASED]++;
    fifo_put(t, js, data, h.len, 0);
    if (!out) return;
    if (out->n >= out->cap || (out->payload && out->pn + h.len > out->pcap)) {
        out->truncated = true;
        return;
    }
    ifh_rtp_rec &r = out->recs[out->n++];
    r.stream = stream;
    r.type = IFH_RTP_FRAME_RTP;
    r.lseq_start = r.lseq_end = h.lseq;
    r.ts = h.ts;
    r.ts_diff = 0;
    r.hdr = h.hdr;
    r.payload_len = h.len;
    r.payload_off = out->payload ? out->pn : -1;
    if (out->payload) {
        memcpy(out->payload + out->pn, data, (size_t)h.len);
        out->pn += h.len;
    }
}

void emit_ers(RtpTable *t, JStream &js, int stream, const Held &head, Sink *out)
{
    const uint32_t ts_diff = head.ts - js.next_ts;          // modulo 2^32, like the timestamps
    const int64_t lo = js.last_lseq + 1, hi = head.lseq - 1;
    js.last_lseq = hi;
    js.next_ts = head.ts;
    js.stats[IFH_RTP_STAT_ERS_EVENTS]++;
    js.stats[IFH_RTP_STAT_ERS_PACKETS] += hi - lo + 1;
    // InfernRTPIngest.py:84-86: rtp_data = codec.silence(ts_diff); G711Codec.silence(n) = fill * n bytes.
    // A timestamp that went backwards would show up as a huge difference: bound it by the FIFO.
    const int64_t nfill = std::min<int64_t>((int64_t)ts_diff / t->ts_per_byte, t->fifo_cap);
    js.stats[IFH_RTP_STAT_ERS_BYTES] += nfill;
    fifo_put(t, js, nullptr, nfill, t->fill);
    if (!out) return;
    if (out->n >= out->cap) {
        out->truncated = true;
        return;
    }
    ifh_rtp_rec &r = out->recs[out->n++];
    memset(&r, 0, sizeof(r));
    r.stream = stream;
    r.type = IFH_RTP_FRAME_ERS;
    r.lseq_start = lo;
    r.lseq_end = hi;
    r.ts = head.ts;
    r.ts_diff = ts_diff;
    r.payload_off = -1;
    r.payload_len = (int32_t)nfill;
}

// release everything at the front of `held` that continues the output sequence
void flush_contiguous(RtpTable *t, JStream &js, int stream, Sink *out)
{
    size_t k = 0;
    while (k < js.held.size() && js.held[k].lseq == js.last_lseq + 1) {
        emit_rtp(t, js, stream, js.held[k], slab_ptr(t, stream, js.held[k].slab), out);
        js.free_slabs.push_back(js.held[k].slab);
        k++;
    }
    js.held.erase(js.held.begin(), js.held.begin() + k);
}

void reset_stream(RtpTable *t, JStream &js)
{
    js.have_ref = js.have_out = false;
    js.held.clear();
    js.free_slabs.clear();
    for (int k = t->depth; k >= 0; --k) js.free_slabs.push_back(k);
}

int push_one(RtpTable *t, int stream, const uint8_t *pkt, int len, Sink *out)
{
    JStream &js = t->st[stream];
    ifh_rtp_hdr hdr;
    const int rc = ifh_rtp_parse(pkt, len, &hdr);
    if (rc != 0) {
        js.stats[IFH_RTP_STAT_PARSE_ERRORS]++;
        return rc;
    }
    if (hdr.payload_len > IFH_RTP_MAX_PAYLOAD) {
        js.stats[IFH_RTP_STAT_PARSE_ERRORS]++;
        return fail(IFH_ERTPPARSE, "ifh_rtpjb_push: payload longer than IFH_RTP_MAX_PAYLOAD");
    }
    js.stats[IFH_RTP_STAT_RECEIVED]++;
    // 16-bit sequence number -> extended ("long") sequence number, nearest to the highest one seen
    int64_t lseq;
    if (!js.have_ref) {
        lseq = hdr.seq;
        js.have_ref = true;
        js.ref_lseq = lseq;
    } else {
        lseq = js.ref_lseq + (int16_t)(uint16_t)(hdr.seq - (uint16_t)(js.ref_lseq & 0xffff));
        if (lseq > js.ref_lseq) js.ref_lseq = lseq;
    }
    const uint8_t *data = pkt + hdr.payload_off;
    Held h{lseq, hdr.ts, hdr.payload_len, -1, hdr};
    if (js.have_out && lseq <= js.last_lseq) {
        js.stats[IFH_RTP_STAT_LATE]++;
        return 0;
    }
    if (!js.have_out || lseq == js.last_lseq + 1) {      // in order (or the very first): straight through
        emit_rtp(t, js, stream, h, data, out);
        flush_contiguous(t, js, stream, out);
        return 0;
    }
    auto pos = std::lower_bound(js.held.begin(), js.held.end(), lseq, [](const Held &a, int64_t v) { return a.lseq < v; });
    if (pos != js.held.end() && pos->lseq == lseq) {
        js.stats[IFH_RTP_STAT_DUPLICATE]++;
        return 0;
    }
    h.slab = js.free_slabs.back();
    js.free_slabs.pop_back();
    memcpy(slab_ptr(t, stream, h.slab), data, (size_t)h.len);
    js.held.insert(pos, h);
    js.stats[IFH_RTP_STAT_REORDERED]++;
    while ((int)js.held.size() > t->depth) {              // the buffer is full: give up on the gap in front
        emit_ers(t, js, stream, js.held.front(), out);
        flush_contiguous(t, js, stream, out);
    }
    return 0;
}

RtpTable *as_table(ifh_rtpjb_t h) { return reinterpret_cast<RtpTable *>(h); }



}  // namespace
}  // namespace ifh

using namespace ifh;

extern "C" {

int ifh_rtp_parse(const uint8_t *pkt, int len, ifh_rtp_hdr *out)
{
    IFH_CHECK_ARG(pkt != nullptr && out != nullptr);
    if (len < 12) return fail(IFH_ERTPPARSE, "ifh_rtp_parse: datagram shorter than the fixed RTP header");
    if ((pkt[0] >> 6) != 2) return fail(IFH_ERTPPARSE, "ifh_rtp_parse: RTP version is not 2");
    const int padding = (pkt[0] >> 5) & 1, ext = (pkt[0] >> 4) & 1, cc = pkt[0] & 15;
    int off = 12 + 4 * cc;
    if (off > len) return fail(IFH_ERTPPARSE, "ifh_rtp_parse: CSRC list runs past the datagram");
    if (ext) {
        if (off + 4 > len) return fail(IFH_ERTPPARSE, "ifh_rtp_parse: header extension runs past the datagram");
        off += 4 + 4 * ((pkt[off + 2] << 8) | pkt[off + 3]);
        if (off > len) return fail(IFH_ERTPPARSE, "ifh_rtp_parse: header extension runs past the datagram");
    }
    int plen = len - off;
    if (padding) {
        const int pad = plen > 0 ? pkt[len - 1] : 0;
        if (pad == 0 || pad > plen) return fail(IFH_ERTPPARSE, "ifh_rtp_parse: bad padding count");
        plen -= pad;
    }
    out->version = 2;
    out->padding = padding;
    out->extension = ext;
    out->cc = cc;
    out->marker = pkt[1] >> 7;
    out->pt = pkt[1] & 0x7f;
    out->seq = (uint16_t)((pkt[2] << 8) | pkt[3]);
    out->ts = ((uint32_t)pkt[4] << 24) | ((uint32_t)pkt[5] << 16) | ((uint32_t)pkt[6] << 8) | pkt[7];
    out->ssrc = ((uint32_t)pkt[8] << 24) | ((uint32_t)pkt[9] << 16) | ((uint32_t)pkt[10] << 8) | pkt[11];
    out->payload_off = off;
    out->payload_len = plen;
    return 0;
}

int ifh_rtpjb_create(int n_streams, int depth, int frame_bytes, int ts_per_byte, int fill_byte, int fifo_cap,
                     ifh_rtpjb_t *out)
{
    IFH_CHECK_ARG(out != nullptr);
    IFH_CHECK_ARG(n_streams >= 1 && depth >= 1 && depth <= 64);
    IFH_CHECK_ARG(frame_bytes >= 1 && ts_per_byte >= 1 && fifo_cap >= 2 * frame_bytes);
    IFH_CHECK_ARG(fill_byte >= 0 && fill_byte <= 255);
    RtpTable *t = new (std::nothrow) RtpTable();
    if (!t) return fail(IFH_ENOMEM, "ifh_rtpjb_create: out of host memory");
    try {
        t->n = n_streams;
        t->depth = depth;
        t->frame_bytes = frame_bytes;
        t->ts_per_byte = ts_per_byte;
        t->fill = fill_byte;
        t->fifo_cap = fifo_cap;
        t->st.resize(n_streams);
        t->slabs.resize((size_t)n_streams * (depth + 1) * IFH_RTP_MAX_PAYLOAD);
        for (auto &js : t->st) {
            js.fifo.resize(fifo_cap);
            js.held.reserve(depth + 1);
            reset_stream(t, js);
        }
    } catch (const std::bad_alloc &) {
        delete t;
        return fail(IFH_ENOMEM, "ifh_rtpjb_create: out of host memory");
    }
    *out = t;
    return 0;
}

int ifh_rtpjb_destroy(ifh_rtpjb_t h)
{
    delete as_table(h);
    return 0;
}

int ifh_rtpjb_reset_stream(ifh_rtpjb_t h, int stream, int drop_fifo)
{
    RtpTable *t = as_table(h);
    IFH_CHECK_ARG(t != nullptr && stream >= 0 && stream < t->n);
    reset_stream(t, t->st[stream]);
    if (drop_fifo) t->st[stream].head = t->st[stream].count = 0;
    return 0;
}

int ifh_rtpjb_push(ifh_rtpjb_t h, int stream, const uint8_t *pkt, int len, ifh_rtp_rec *recs, int rec_cap,
                   uint8_t *payload, int64_t payload_cap, int *nrec)
{
    RtpTable *t = as_table(h);
    IFH_CHECK_ARG(t != nullptr && stream >= 0 && stream < t->n && pkt != nullptr);
    IFH_CHECK_ARG(rec_cap == 0 || recs != nullptr);
    Sink sink{recs, rec_cap, 0, payload, payload_cap, 0, false};
    const int rc = push_one(t, stream, pkt, len, recs ? &sink : nullptr);
    if (nrec) *nrec = sink.n;
    if (rc != 0) return rc;
    if (sink.truncated) return fail(IFH_EINVAL, "ifh_rtpjb_push: record or payload buffer too small for what was released");
    return 0;
}

int ifh_rtpjb_push_batch(ifh_rtpjb_t h, const uint8_t *buf, const int32_t *off, const int32_t *stream, int n,
                         int32_t *status)
{
    RtpTable *t = as_table(h);
    IFH_CHECK_ARG(t != nullptr && n >= 0 && (n == 0 || (buf != nullptr && off != nullptr && stream != nullptr)));
    int bad = 0;
    for (int i = 0; i < n; ++i) {
        int rc;
        if (stream[i] < 0 || stream[i] >= t->n || off[i + 1] < off[i]) rc = IFH_EINVAL;
        else rc = push_one(t, stream[i], buf + off[i], off[i + 1] - off[i], nullptr);
        if (status) status[i] = rc;
        bad += rc != 0;
    }
    return bad;         // number of datagrams refused (0 = all taken); per-datagram codes in status[]
}

int ifh_rtpjb_pop_tick(ifh_rtpjb_t h, uint8_t *frames, int32_t *slots, int cap, int *n_out)
{
    RtpTable *t = as_table(h);
    IFH_CHECK_ARG(t != nullptr && frames != nullptr && slots != nullptr && n_out != nullptr && cap >= 0);
    int n = 0;
    const int fb = t->frame_bytes;
    for (int s = 0; s < t->n && n < cap; ++s) {
        JStream &js = t->st[s];
        if (js.count < fb) continue;
        uint8_t *dst = frames + (size_t)n * fb;
        const int64_t run = std::min<int64_t>(fb, t->fifo_cap - js.head);
        memcpy(dst, js.fifo.data() + js.head, (size_t)run);
        if (run < fb) memcpy(dst + run, js.fifo.data(), (size_t)(fb - run));
        js.head = (js.head + fb) % t->fifo_cap;
        js.count -= fb;
        slots[n++] = s;
    }
    *n_out = n;
    return 0;
}

int ifh_rtpjb_stats(ifh_rtpjb_t h, int stream, int64_t *stats)
{
    RtpTable *t = as_table(h);
    IFH_CHECK_ARG(t != nullptr && stream >= 0 && stream < t->n && stats != nullptr);
    JStream &js = t->st[stream];
    memcpy(stats, js.stats, sizeof(js.stats));
    stats[IFH_RTP_STAT_FIFO_BYTES] = js.count;
    stats[IFH_RTP_STAT_HELD] = (int64_t)js.held.size();
    stats[IFH_RTP_STAT_LAST_LSEQ] = js.have_out ? js.last_lseq : -1;
    return 0;
}

}  // extern "C"
