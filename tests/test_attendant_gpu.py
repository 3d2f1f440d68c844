"""GPU: the three actors of the AI-attendant path (BASELINE configuration 5) wired the way Apps/AIAttendant/AIASession.py
wires them -- STT result -> LLM session text in -> sentence pieces out -> TTS say -> audio chunks to the call's sound
output -- on seeded weights, through the reference's actor / session method names only."""
import threading

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


class WhisperStubTokenizer:
    eos_token_id = 50257
    ids = {'<|startoftranscript|>': 50258, '<|en|>': 50259, '<|transcribe|>': 50359, '<|translate|>': 50358,
           '<|notimestamps|>': 50363, '<|nospeech|>': 50362}

    def convert_tokens_to_ids(self, t):
        return self.ids[t] if isinstance(t, str) else [self.ids[x] for x in t]

    def decode(self, ids, skip_special_tokens=True):
        return ' ' + ' '.join('word%d' % (i % 97) for i in ids)


class CharIdsProcessor:
    """TTS text front end stand-in: one SpeechT5 character id per character"""

    def __call__(self, text, return_tensors='pt'):
        ids = [4 + (ord(c) % 70) for c in text][:60] or [5]
        return {'input_ids': torch.tensor([ids], dtype=torch.long)}


def test_attendant_turn_through_the_three_actors(built_lib):
    from infernos_amd import _lib
    from infernos_amd.actors import InfernLLMActor, InfernSTTActor, InfernTTSActor
    from infernos_amd.audio import AudioChunk, VadAudioChunk
    from infernos_amd.llm import LLMRequest, LLMResult, LLMSessionParams
    from infernos_amd.muxer import ASMarkerNewSent
    from infernos_amd.stt import STTRequest, STTResult
    from infernos_amd.synth import CharChatTokenizer, synth_utterance
    from infernos_amd.tts import TTSRequest
    from infernos_amd.weights import QWEN2_CONFIGS, synth_state_dict
    dev = _lib.require_device('cuda:0')
    cfg = QWEN2_CONFIGS['qwen2_tiny64']
    g = torch.Generator().manual_seed(1)
    voices = [torch.randn(1, 512, generator=g) for _ in range(4)]
    stt = InfernSTTActor(weights=synth_state_dict('whisper_tiny', 0), tokenizer=WhisperStubTokenizer(), fixed_new_tokens=5)
    llm = InfernLLMActor(weights=synth_state_dict('qwen2_tiny64', 1), config=cfg, tokenizer=CharChatTokenizer(cfg['vocab']),
                         max_new_tokens=40, max_tokens=512)
    tts = InfernTTSActor(weights={'speecht5_tts': synth_state_dict('speecht5_tts', 0), 'hifigan': synth_state_dict('hifigan', 0),
                                  'amendment': synth_state_dict('amendment', 0)}, processor=CharIdsProcessor(),
                         speaker_embeddings=voices)
    stt.start(dev)
    llm.start(dev, warmup=False)
    tts.start('en', 8000, dev)
    try:
        ncalls = 3
        sound = {i: [] for i in range(ncalls)}
        heard, said = {}, {i: [] for i in range(ncalls)}
        done = threading.Event()
        lock = threading.Lock()
        sess = []
        for i in range(ncalls):
            s = dict(stt=stt.new_stt_session(), llm=llm.new_llm_session(LLMSessionParams('You are attendant %d.' % i)),
                     tts=tts.new_tts_session())
            tts.tts_session_start(s['tts'], (lambda chunk, i=i: on_sound(i, chunk)))
            sess.append(s)

        def on_sound(i, chunk):
            with lock:
                sound[i].append(chunk)
                if all(any(isinstance(c, ASMarkerNewSent) for c in sound[k]) for k in range(ncalls)):
                    done.set()

        def on_llm(i, result):
            assert isinstance(result, LLMResult)
            said[i].append(result.text)
            tts.tts_session_say(sess[i]['tts'], TTSRequest(result.text, speaker_id=i))

        def on_stt(i, result):
            assert isinstance(result, STTResult)
            heard[i] = result.text
            req = LLMRequest(result.text, (lambda result, i=i: on_llm(i, result)))
            llm.llm_session_textin(sess[i]['llm'], req)

        for i in range(ncalls):
            x = torch.from_numpy(synth_utterance(1000 + i, 4.0)[8000:24000]).to(dev)
            stt.stt_session_soundin(sess[i]['stt'], STTRequest(VadAudioChunk(x, 8000, 8000), (lambda result, i=i: on_stt(i, result)), 'en'))
        assert done.wait(240), (heard, said, {k: len(v) for k, v in sound.items()})
        for i in range(ncalls):
            assert heard[i].startswith('word') and len(heard[i].split()) == 5
            assert said[i] and all(isinstance(t, str) and t for t in said[i])
            ctx = llm.sessions[sess[i]['llm']].context
            assert ctx[0] == {'role': 'system', 'content': 'You are attendant %d.' % i}
            assert ctx[1] == {'role': 'user', 'content': heard[i]}
            assert ctx[2]['role'] == 'assistant' and ctx[2]['content'].split(' ')[0] == said[i][0].split(' ')[0]
            chunks = [c for c in sound[i] if isinstance(c, AudioChunk)]
            assert chunks and all(c.samplerate == 8000 and not c.audio.is_cuda and torch.isfinite(c.audio.float()).all() for c in chunks)
            assert sum(c.audio.numel() for c in chunks) >= 2048
        for s in sess:
            stt.stt_session_end(s['stt'])
            llm.llm_session_end(s['llm'])
            tts.tts_session_end(s['tts'])
        assert not stt.sessions and not llm.sessions and not tts.sessions
    finally:
        stt.stop()
        llm.stop()
        tts.stop()
