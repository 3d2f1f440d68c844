"""ctypes binding of libinfernos_hip.so (include/infernos_hip.h).

The product has no CPU fallback: if the shared library is missing, or a compute entry
point is reached without a HIP device, a RuntimeError is raised.
"""
import ctypes
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libinfernos_hip.so')
if os.environ.get('IFH_LIB_PATH'):          # tools/ only: an experimental build of the same library
    LIB_PATH = os.environ['IFH_LIB_PATH']

c_i32p = ctypes.POINTER(ctypes.c_int32)
_vp, _i, _i64, _f, _d = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_double

# name -> (restype, argtypes); mirrors include/infernos_hip.h one to one
SIGNATURES = {
    'ifh_last_error': (ctypes.c_char_p, []),
    'ifh_version': (_i, []),
    'ifh_device_count': (_i, []),
    'ifh_g711_tables_host': (_i, [_vp, _vp]),
    'ifh_g711_decode_u8_f32': (_i, [_vp, _vp, _i64, _vp]),
    'ifh_g711_encode_f32_u8': (_i, [_vp, _vp, _i64, _vp]),
    'ifh_resample_create': (_i, [_i, _i, ctypes.POINTER(_vp)]),
    'ifh_resample_destroy': (_i, [_vp]),
    'ifh_resample_info': (_i, [_vp, c_i32p, c_i32p, c_i32p, c_i32p, _vp]),
    'ifh_resample_out_len': (_i64, [_vp, _i64]),
    'ifh_resample_run': (_i, [_vp, _vp, _i64, _vp, _i64, _i, _vp, _i64, _vp]),
    'ifh_ingest_tick': (_i, [_vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'ifh_mux_encode_f32_u8': (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    'ifh_rtp_parse': (_i, [_vp, _i, _vp]),
    'ifh_rtpjb_create': (_i, [_i, _i, _i, _i, _i, _i, ctypes.POINTER(_vp)]),
    'ifh_rtpjb_destroy': (_i, [_vp]),
    'ifh_rtpjb_reset_stream': (_i, [_vp, _i, _i]),
    'ifh_rtpjb_push': (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _i64, c_i32p]),
    'ifh_rtpjb_push_batch': (_i, [_vp, _vp, _vp, _vp, _i, _vp]),
    'ifh_rtpjb_pop_tick': (_i, [_vp, _vp, _vp, _i, c_i32p]),
    'ifh_rtpjb_stats': (_i, [_vp, _i, _vp]),
    'ifh_vad_energy_prob': (_i, [_vp, _vp, _i, _vp, _vp]),
    'ifh_vadnet_weight_floats': (_i, []),
    'ifh_vadnet_prob': (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'ifh_vad_fsm_step': (_i, [_vp, _vp, _i, _i, _i, _d, _vp, _vp, _vp]),
    'ifh_vad_step': (_i, [_vp, _vp, _vp, _i, _i, _d, _vp, _vp, _vp, _vp, _vp, _vp]),
    'ifh_ingest_block': (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _d, _vp, _vp, _vp, _vp, _vp,
                              _vp, _i64, _vp, _i, _vp, _vp, _vp]),
    'ifh_ingest_block_net': (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _d, _vp, _vp, _vp, _vp, _vp,
                                  _vp, _i64, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    'ifh_logmel_create': (_i, [_i, ctypes.POINTER(_vp)]),
    'ifh_logmel_destroy': (_i, [_vp]),
    'ifh_logmel_filters_host': (_i, [_vp, _vp]),
    'ifh_logmel_workspace_floats': (_i64, [_vp, _i, _i]),
    'ifh_logmel_run': (_i, [_vp, _vp, _i64, _vp, _i, _vp, _i, _vp, _vp]),
    'ifh_logmel_run_raw': (_i, [_vp, _vp, _i64, _vp, _i, _vp, _vp, _vp]),
    'ifh_logmel_finish_transpose_bf16': (_i, [_vp, _vp, _vp, _i, _vp, _vp]),
}



class ConvDesc(ctypes.Structure):
    """ifh_conv_desc (include/infernos_hip.h)"""
    _fields_ = [('x', _vp), ('x_bstride', _i64), ('lda', ctypes.c_int32),
                ('cin', ctypes.c_int32), ('taps', ctypes.c_int32), ('stride', ctypes.c_int32), ('dil', ctypes.c_int32),
                ('pad', ctypes.c_int32), ('t_in', ctypes.c_int32), ('t_out', ctypes.c_int32), ('nbatch', ctypes.c_int32),
                ('w', _vp), ('n', ctypes.c_int32), ('bias', _vp), ('colmask', _vp), ('pre_slope', _f),
                ('act', ctypes.c_int32), ('act_slope', _f), ('resid', _vp), ('resid_bstride', _i64),
                ('resid_ld', ctypes.c_int32), ('out_scale', _f), ('accumulate', ctypes.c_int32), ('out', _vp),
                ('out_f32', ctypes.c_int32), ('out_bstride', _i64), ('ldc', ctypes.c_int32), ('ostride', ctypes.c_int32),
                ('ooff', ctypes.c_int32), ('dyn_pos', _vp), ('dyn_ooff_mul', ctypes.c_int32), ('dyn_resid_mul', _i64),
                ('n_split', ctypes.c_int32), ('out2', _vp), ('out2_bstride', _i64), ('ldc2', ctypes.c_int32),
                ('ooff2', ctypes.c_int32), ('dyn_ooff2_mul', ctypes.c_int32),
                ('aln_stats', _vp), ('aln_c1', _vp), ('rln_stats', _vp), ('rln_gamma', _vp), ('rln_beta', _vp),
                ('stats_out', _vp), ('ln_dim', ctypes.c_int32), ('ln_eps', _f), ('ln_rms', ctypes.c_int32),
                ('dyn_stride', ctypes.c_int32), ('decode_step', ctypes.c_int32), ('convt_cout', ctypes.c_int32),
                ('splitk_ws', _vp), ('splitk_ws_floats', _i64), ('argmax_keys', _vp), ('whole_chip', ctypes.c_int32)]


class ResblockDesc(ctypes.Structure):
    """ifh_resblock_desc (include/infernos_hip.h)"""
    _fields_ = [('x', _vp), ('x_bstride', _i64), ('c', ctypes.c_int32), ('taps', ctypes.c_int32), ('dil', ctypes.c_int32),
                ('t', ctypes.c_int32), ('nbatch', ctypes.c_int32), ('w1', _vp), ('bias1', _vp), ('w2', _vp), ('bias2', _vp),
                ('slope', _f), ('out_scale', _f), ('accumulate', ctypes.c_int32), ('out', _vp), ('out_bstride', _i64)]


class ChainDesc(ctypes.Structure):
    """ifh_chain_desc (include/infernos_hip.h)"""
    _fields_ = [('x', _vp), ('x_bstride', _i64), ('c', ctypes.c_int32), ('taps', ctypes.c_int32), ('t', ctypes.c_int32),
                ('nbatch', ctypes.c_int32), ('wstream', _vp), ('nunits', ctypes.c_int32), ('bias', _vp), ('slope', _f),
                ('out_scale', _f), ('accumulate', ctypes.c_int32), ('out', _vp), ('out_bstride', _i64), ('debug_prof', _vp),
                ('post_slope', _f)]


class SeqDesc(ctypes.Structure):
    """ifh_seq_desc (include/infernos_hip.h)"""
    _fields_ = [('x', _vp), ('x_bstride', _i64), ('c', ctypes.c_int32), ('taps', ctypes.c_int32), ('t', ctypes.c_int32),
                ('nbatch', ctypes.c_int32), ('wstream', _vp), ('nunits', ctypes.c_int32), ('bias', _vp), ('slope', _f),
                ('out_scale', _f), ('accumulate', ctypes.c_int32), ('out', _vp), ('out_bstride', _i64), ('debug_prof', _vp),
                ('post_slope', _f)]


class LevelDesc(ctypes.Structure):
    """ifh_level_desc (include/infernos_hip.h)"""
    _fields_ = [('x', _vp), ('x_bstride', _i64), ('c', ctypes.c_int32), ('t', ctypes.c_int32), ('nbatch', ctypes.c_int32),
                ('nblocks', ctypes.c_int32), ('taps', ctypes.c_int32 * 3), ('accumulate', ctypes.c_int32),
                ('wstream', _vp * 3), ('bias', _vp * 3), ('slope', _f), ('out_scale', _f), ('out', _vp), ('out_bstride', _i64),
                ('debug_prof', _vp), ('post_w', _vp), ('post_bias', _f), ('post_slope', _f), ('audio', _vp), ('mean_ws', _vp),
                ('mean_ws_bytes', _i64)]


class Ring256Desc(ctypes.Structure):
    """ifh_ring256_desc (include/infernos_hip.h)"""
    _fields_ = [('x', _vp), ('x_bstride', _i64), ('taps', ctypes.c_int32), ('dil', ctypes.c_int32), ('t', ctypes.c_int32),
                ('nbatch', ctypes.c_int32), ('wstream', _vp), ('bias', _vp), ('pre_slope', _f), ('resid', _vp),
                ('resid_bstride', _i64), ('out_scale', _f), ('accumulate', ctypes.c_int32), ('out', _vp), ('out_bstride', _i64)]


class BeamDesc(ctypes.Structure):
    """ifh_beam_desc (include/infernos_hip.h)"""
    _fields_ = [('logits', _vp), ('ld', _i64), ('vocab', ctypes.c_int32), ('nbatch', ctypes.c_int32), ('beams', ctypes.c_int32),
                ('suppress', _vp), ('begin_suppress', _vp), ('toks', _vp), ('pos', _vp), ('prompt_len', ctypes.c_int32),
                ('max_length', ctypes.c_int32), ('eos_id', ctypes.c_int32), ('length_penalty', _f), ('run_scores', _vp),
                ('fin_scores', _vp), ('fin_seqs', _vp), ('fin_len', _vp), ('is_fin', _vp), ('unsat', _vp), ('beam_src', _vp),
                ('alive', _vp), ('scratch', _vp)]


class GqaDesc(ctypes.Structure):
    """ifh_gqa_desc (include/infernos_hip.h)"""
    _fields_ = [('q', _vp), ('q_ts', _i64), ('cache', _vp), ('cache_bs', _i64), ('cache_ts', _i64), ('v_off', ctypes.c_int32),
                ('out', _vp), ('o_ts', _i64), ('key_len', _vp), ('ntokens', ctypes.c_int32), ('tokens_per_row', ctypes.c_int32),
                ('nheads', ctypes.c_int32), ('nkv', ctypes.c_int32), ('head_dim', ctypes.c_int32), ('max_keys', ctypes.c_int32),
                ('scale', _f), ('rope_cos_sin', _vp)]


class RtpHdr(ctypes.Structure):
    """ifh_rtp_hdr (include/infernos_hip.h)"""
    _fields_ = [(n, ctypes.c_int32) for n in ('version', 'padding', 'extension', 'cc', 'marker', 'pt')] + \
               [(n, ctypes.c_uint32) for n in ('seq', 'ts', 'ssrc')] + \
               [('payload_off', ctypes.c_int32), ('payload_len', ctypes.c_int32)]


class RtpRec(ctypes.Structure):
    """ifh_rtp_rec (include/infernos_hip.h)"""
    _fields_ = [('stream', ctypes.c_int32), ('type', ctypes.c_int32), ('lseq_start', _i64), ('lseq_end', _i64),
                ('ts', ctypes.c_uint32), ('ts_diff', ctypes.c_uint32), ('payload_off', _i64),
                ('payload_len', ctypes.c_int32), ('hdr', RtpHdr)]


RTP_STATS = ('received', 'released', 'late', 'duplicate', 'reordered', 'ers_events', 'ers_packets', 'ers_bytes',
             'parse_errors', 'overflow_bytes', 'fifo_bytes', 'held', 'last_lseq')
IFH_ERTPPARSE = -4
IFH_RTP_MAX_PAYLOAD = 1472


class AttnDesc(ctypes.Structure):
    """ifh_attn_desc (include/infernos_hip.h)"""
    _fields_ = [('q', _vp), ('k', _vp), ('v', _vp), ('out', _vp),
                ('q_bs', _i64), ('q_ts', _i64), ('k_bs', _i64), ('k_ts', _i64), ('v_bs', _i64), ('v_ts', _i64),
                ('o_bs', _i64), ('o_ts', _i64),
                ('nbatch', ctypes.c_int32), ('nheads', ctypes.c_int32), ('head_dim', ctypes.c_int32),
                ('tq', ctypes.c_int32), ('tk', ctypes.c_int32), ('key_len', _vp), ('relbias', _vp), ('nrel', ctypes.c_int32)]


SIGNATURES.update({
    'ifh_conv_bf16': (_i, [ctypes.POINTER(ConvDesc), _vp]),
    'ifh_resblock_pair_bf16': (_i, [ctypes.POINTER(ResblockDesc), _vp]),
    'ifh_resblock_chain_bf16': (_i, [ctypes.POINTER(ChainDesc), _vp]),
    'ifh_resblock_seq_bf16': (_i, [ctypes.POINTER(SeqDesc), _vp]),
    'ifh_resblock_seq_unit_bytes': (_i, [_i]),
    'ifh_resblock_seq_supported': (_i, [_i, _i, _i]),
    'ifh_conv_ring256_bf16': (_i, [ctypes.POINTER(Ring256Desc), _vp]),
    'ifh_resblock_level_bf16': (_i, [ctypes.POINTER(LevelDesc), _vp]),
    'ifh_level_ws_bytes': (_i64, []),
    'ifh_layernorm_bf16': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _f, _vp]),
    'ifh_transpose_to_bf16': (_i, [_vp, _i, _vp, _i, _i, _i, _vp]),
    'ifh_attn_prefill_bf16': (_i, [ctypes.POINTER(AttnDesc), _vp]),
    'ifh_attn_decode_bf16': (_i, [_vp, _i64, _vp, _vp, _i64, _i64, _vp, _i64, _vp, _i, _i, _i, _i, _vp, _i, _vp]),
    'ifh_embed_bf16': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _vp]),
    'ifh_add_i32': (_i, [_vp, _i, _vp, _i64, _vp]),
    'ifh_attn_decode_shared_bf16': (_i, [_vp, _i64, _vp, _vp, _i64, _i64, _vp, _i64, _i, _i, _i, _i, _i, _vp]),
    'ifh_beam_step': (_i, [ctypes.POINTER(BeamDesc), _vp]),
    'ifh_kv_gather_bf16': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i64, _i, _i, _i64, _vp]),
    'ifh_rmsnorm_bf16': (_i, [_vp, _vp, _vp, _i, _i, _f, _vp]),
    'ifh_rope_append_bf16': (_i, [_vp, _i64, _vp, _i, _vp, _i64, _i64, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    'ifh_attn_gqa_bf16': (_i, [ctypes.POINTER(GqaDesc), _vp]),
    'ifh_silu_mul_bf16': (_i, [_vp, _vp, _i64, _i, _i, _vp]),
    'ifh_add_i32_vec': (_i, [_vp, _vp, _i, _i, _vp]),
    'ifh_repetition_penalty_f32': (_i, [_vp, _i64, _i, _i, _vp, _i64, _vp, _f, _vp]),
    'ifh_sample_topk_f32': (_i, [_vp, _i64, _i, _i, _f, _i, _f, _vp, _vp, _vp, _vp, _vp, _vp]),
    'ifh_g722_init': (_i, [_vp, _i, _vp]),
    'ifh_g722_encode': (_i, [_vp, _vp, _i, _i64, _i, _i, _vp, _i64, _i, _vp]),
    'ifh_g722_decode': (_i, [_vp, _vp, _i64, _i, _i, _vp, _i, _i64, _i, _vp]),
    'ifh_argmax_pick_f32': (_i, [_vp, _i64, _i, _i, _i, _vp, _vp, _vp, _i, _vp]),
    'ifh_conv_argmax_supported': (_i, [_i, _i, _i]),
    'ifh_argmax_keys_finish': (_i, [_vp, _vp, _i, _vp]),
    'ifh_tts_stop_update': (_i, [_vp, _vp, _i, _i, _i, _i, _f, _i, _vp, _i, _vp, _vp]),
    'ifh_tts_stop_advance': (_i, [_vp, _vp, _i, _i, _i, _f, _i, _vp, _i, _vp, _i64, _vp, _vp]),
    'ifh_tts_chunks_bf16': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    'ifh_tts_chunks_rows_bf16': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    'ifh_tts_stop_advance_rows': (_i, [_vp, _vp, _i, _f, _i, _vp, _vp, _vp, _i, _vp, _i64, _vp]),
    'ifh_tts_carry_rows_bf16': (_i, [_vp, _vp, _vp, _i, _i, _vp]),
    'ifh_hifigan_post_bf16': (_i, [_vp, _vp, _f, _vp, _i, _i, _f, _vp]),
    'ifh_amend_final_bf16': (_i, [_vp, _vp, _vp, _i, _vp]),
    'ifh_l2norm_rows_bf16': (_i, [_vp, _i, _i, _vp, _i, _vp]),
    'ifh_stream_create_cu_range': (_i, [_i, _i, ctypes.POINTER(ctypes.c_void_p)]),
    'ifh_stream_destroy': (_i, [_vp]),
    'ifh_set_cu_budget': (_i, [_i]),
})

_lib = None
_lock = threading.Lock()
# Statistic: calls into stream-taking entry points (~ kernel launches issued through the C ABI), including those replayed from
# captured hipGraphs (CountedGraph); bench.py reports it per utterance cycle.
CALLS = [0]
_HOST_ONLY = ('ifh_g711_tables_host',)


_tls = threading.local()


def _counted(fn):
    def call(*a):
        cap = getattr(_tls, 'cap', None)
        if cap is None:
            CALLS[0] += 1
        else:                     # this thread is capturing a hipGraph: the launch is recorded, not run -- counted per replay
            cap[0] += 1
        return fn(*a)
    call.__name__ = getattr(fn, '__name__', 'ifh')
    return call


# Captures and graph destruction never overlap: torch's ~CUDAGraph ends with a device-wide synchronisation (ROCm >= 6.2), which HIP
# refuses while a stream of the process is capturing -- inside a destructor that is std::terminate, i.e. abort() of the whole serving
# process, with no message (round 6: a dead pipeline's graphs reached the garbage collector on the main thread while an engine
# thread of the live one was capturing a decoder step; found under rocgdb).  So a CountedGraph never lets its graph die where the
# collector happens to find it: __del__ parks it, and parked graphs are destroyed under the lock every capture takes.
_graph_lock = threading.RLock()
_graveyard = []


def _drain_graveyard():
    """destroy parked graphs; the caller holds _graph_lock"""
    while _graveyard:
        g = _graveyard.pop()
        del g


def release_graphs():
    """Destroy the hipGraphs of CountedGraph objects that are no longer referenced (SpeechPipeline.close() calls this).  Blocks while
    another thread captures."""
    with _graph_lock:
        _drain_graveyard()


class CountedGraph:
    """hipGraph of the launches `fn` issues on the current stream (thread-local capture), remembering how many C-ABI calls
    it holds so that replays keep the CALLS statistic meaningful.  The count is this thread's own: launches other threads
    issue during the capture stay in the running total and out of the graph's number."""

    captures = [0]               # statistic: graphs captured so far (a capture inside a serving loop synchronises the device)

    def __init__(self, fn):
        import torch
        with _graph_lock:
            _drain_graveyard()
            CountedGraph.captures[0] += 1
            torch.cuda.synchronize()
            self.g = torch.cuda.CUDAGraph()
            prev, cap = getattr(_tls, 'cap', None), [0]
            _tls.cap = cap
            try:
                with torch.cuda.graph(self.g, capture_error_mode='thread_local'):      # records the launches; nothing executes until replay
                    fn()
            finally:
                _tls.cap = prev
            self.n = cap[0]

    def replay(self):
        self.g.replay()
        CALLS[0] += self.n

    def __del__(self):
        try:
            g = self.__dict__.pop('g', None)
            if g is not None:
                _graveyard.append(g)
        except Exception:           # interpreter shutdown: the module's globals may be gone
            pass


class InfernosHipError(RuntimeError):
    """A failed library call.  `code` is the entry point's return value (IFH_EINVAL = -1: the call was declined before anything was
    launched; IFH_EHIP = -2: a HIP error), None where no call was made."""

    def __init__(self, msg, code=None):
        super().__init__(msg)
        self.code = code


IFH_EINVAL, IFH_EHIP = -1, -2


def lib():
    """The loaded shared library; raises if it has not been built."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise InfernosHipError(
                        'libinfernos_hip.so is missing (%s): run `python -c "import __graft_entry__ as g; g.build()"`; '
                        'there is no CPU fallback' % LIB_PATH)
                # torch bundles its own libamdhip64 (same SONAME as /opt/rocm's).  Import torch
                # first so that ONE HIP runtime is shared by torch (memory, streams) and the
                # kernels here; two runtimes in a process do not see each other's context.
                import torch  # noqa: F401
                L = ctypes.CDLL(LIB_PATH)
                for name, (res, args) in SIGNATURES.items():
                    fn = getattr(L, name)
                    fn.restype = res
                    fn.argtypes = args
                    if args and args[-1] is _vp and name not in _HOST_ONLY:     # stream-taking entry points launch kernels
                        setattr(L, name, _counted(fn))
                _lib = L
    return _lib


def check(rc, what=''):
    if rc != 0:
        msg = lib().ifh_last_error()
        raise InfernosHipError('%s failed (%d): %s' % (what or 'libinfernos_hip', rc, (msg or b'').decode()), code=rc)
    return rc


def require_device(device=None):
    """torch.device of the HIP GPU to run on; raises when there is none."""
    import torch
    if not torch.cuda.is_available():
        raise InfernosHipError('infernos_amd: no HIP device is visible; the speech path has no CPU fallback')
    if device is None or str(device) in ('cuda', 'hip'):
        return torch.device('cuda', torch.cuda.current_device())
    d = torch.device(device)
    if d.type != 'cuda':
        raise InfernosHipError('infernos_amd: device %r is not a HIP GPU; the speech path has no CPU fallback' % (device,))
    return d


_CU_STREAMS = []            # (handle, ExternalStream): CU-range streams live as long as the process


def cu_range_stream(device, first_cu, n_cus):
    """torch stream whose kernels run on CUs [first_cu, first_cu + n_cus) only (ifh_stream_create_cu_range)."""
    import torch
    h = ctypes.c_void_p()
    with torch.cuda.device(device):
        check(lib().ifh_stream_create_cu_range(int(first_cu), int(n_cus), ctypes.byref(h)), 'ifh_stream_create_cu_range')
    s = torch.cuda.ExternalStream(h.value, device=device)
    _CU_STREAMS.append((h, s))
    return s


def throughput_stream(device, priority=0):
    """Stream for a stage of chip-filling kernels.  IFH_BIG_CUS=n (tuning switch, default off) keeps those stages on the first n
    CUs, so that the decode chains (ordinary streams) always find room on the others; the persistent kernels then size their grids
    to n (ifh_set_cu_budget)."""
    import os
    import torch
    n = int(os.environ.get('IFH_BIG_CUS', '0'))
    if n <= 0:
        return torch.cuda.Stream(device=device, priority=priority)
    check(lib().ifh_set_cu_budget(n), 'ifh_set_cu_budget')
    return cu_range_stream(device, 0, n)


def stream_ptr(device=None):
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def ptr(t):
    """Raw device (or host) pointer of a contiguous torch tensor / None."""
    if t is None:
        return None
    assert t.is_contiguous(), 'tensor must be contiguous'
    return ctypes.c_void_p(t.data_ptr())
