"""Register infernos_amd's classes under the reference's own module paths, so that the
unmodified SIP/RTP/app code of Infernos (`from Core.Codecs.G711 import G711Codec`,
`from Cluster.InfernSTTWorker import InfernSTTWorker`, ...) picks up the MI355X path.

    import infernos_amd.compat as compat; compat.install()      # before importing Infernos modules
"""
import sys
import types

MAP = {
    'Core.Codecs.GenCodec': ('infernos_amd.codecs', ['GenCodec']),
    'Core.Codecs.G711': ('infernos_amd.codecs', ['G711Codec']),
    'Core.Codecs.G722': ('infernos_amd.codecs', ['G722Codec']),
    'Core.T2T.Translator': ('infernos_amd.t2t', ['Translator']),
    'Core.T2T.NumbersToWords': ('infernos_amd.t2t', ['NumbersToWords']),
    'Core.AudioChunk': ('infernos_amd.audio', ['AudioChunk', 'VadAudioChunk']),
    'Core.AStreamMarkers': ('infernos_amd.muxer', ['ASMarkerGeneric', 'ASMarkerNewSent', 'ASMarkerSentDoneCB']),
    'Core.OutputMuxer': ('infernos_amd.muxer', ['OutputMuxer', 'OutputMTMuxer']),
    'Core.InfernWrkThread': ('infernos_amd.workers', ['InfernWrkThread', 'RTPWrkTInit', 'RTPWrkTRun', 'RTPWrkTStop']),
    'Core.VAD.SileroVAD': ('infernos_amd.vad', ['VADChannel', 'SileroVADWorker']),
    'Core.VAD.SileroVADUtils': ('infernos_amd.vad', ['VADIteratorB', 'VADChannelState', 'VADBatchState', 'VADBatchFromList']),
    'Cluster.InfernBatchedWorker': ('infernos_amd.workers', ['InfernBatchedWorker']),
    'Cluster.STTSession': ('infernos_amd.stt', ['STTRequest', 'STTSentinel', 'STTResult', 'STTSession']),
    'Cluster.InfernSTTWorker': ('infernos_amd.stt', ['InfernSTTWorker']),
    'Cluster.InfernTTSWorker': ('infernos_amd.tts', ['InfernTTSWorker', 'cleanup_text_eu', 'lang2model']),
    'Cluster.TTSSession': ('infernos_amd.tts', ['TTSRequest', 'TTSSndDispatch', 'TTSSession']),
    'Cluster.LLMSession': ('infernos_amd.llm', ['LLMRequest', 'LLMResult', 'LLMInferRequest', 'LLMSessionParams', 'LLMSession']),
    'Cluster.InfernLLMWorker': ('infernos_amd.llm', ['InfernLLMWorker', 'ResultsStreamer']),
    'HelloSippyTTSRT.HelloSippyRTPipe': ('infernos_amd.tts', ['HelloSippyRTPipe', 'HelloSippyPlayRequest', 'HelloSippyPipeState',
                                                              'HelloSippyPipeStateBatched']),
    'safetorch.InfernTorcher': ('infernos_amd.torcher', ['InfernTorcher', 'InfernTorcherDeadlock', 'rc_filter']),
    'config.InfernGlobals': ('infernos_amd.torcher', ['InfernGlobals']),
    'rtpsynth.RtpJBuf': ('infernos_amd.rtp', ['RtpJBuf', 'RTPFrameType', 'RTPParseError']),
}


def install(override=True):
    import importlib
    for ref_name, (mod_name, names) in MAP.items():
        if ref_name in sys.modules and not override:
            continue
        src = importlib.import_module(mod_name)
        m = types.ModuleType(ref_name)
        m.__dict__.update({n: getattr(src, n) for n in names})
        m.__infernos_amd__ = True
        sys.modules[ref_name] = m
        parent, _, leaf = ref_name.rpartition('.')
        while parent:
            pm = sys.modules.get(parent)
            if pm is None:
                pm = types.ModuleType(parent)
                pm.__path__ = []
                sys.modules[parent] = pm
            setattr(pm, leaf, sys.modules[parent + '.' + leaf])
            parent, _, leaf = parent.rpartition('.')
