"""infernos_amd -- MI355X-native implementation of the Infernos per-call speech path.

Host code mirrors the reference's plugin interface (class and method names, argument
meaning, callback and error behaviour) and calls hand-written gfx950
kernels through the C ABI in include/infernos_hip.h.  Module map (reference file ->
module here):

    Core/Codecs/{GenCodec,G711}.py            -> infernos_amd.codecs
    Core/AudioChunk.py, config/InfernGlobals  -> infernos_amd.audio
    Core/AStreamMarkers.py, Core/OutputMuxer  -> infernos_amd.muxer
    Core/InfernWrkThread.py, Cluster/InfernBatchedWorker.py -> infernos_amd.workers
    Core/VAD/SileroVAD{,Utils}.py             -> infernos_amd.vad
    Cluster/STTSession.py, InfernSTTWorker.py -> infernos_amd.stt
    Cluster/TTSSession.py, InfernTTSWorker.py, HelloSippyTTSRT/HelloSippyRTPipe.py -> infernos_amd.tts
    safetorch/InfernTorcher.py                -> infernos_amd.torcher

infernos_amd.compat.install() registers these under the reference's own module paths so
that the unmodified SIP/RTP orchestration imports them.
"""
__version__ = '0.1.0'
