"""CPU: the text stage (Core/T2T/Translator.py, Core/T2T/NumbersToWords.py) against transcripts of the reference classes
run in place over scripted stand-ins for argostranslate / inflect (tests/golden/t2t.json, tools/gen_golden.py:gen_t2t)."""
import json
import os
import types

import pytest

from infernos_amd.t2t import NumbersToWords, Translator, english_number_to_words


class ScriptedBackend:
    def __init__(self, pairs):
        self.pairs, self.log = [tuple(p) for p in pairs], []

    def load_pair(self, a, b):
        if (a, b) not in self.pairs:
            raise StopIteration
        self.log.append('install %s_%s.argosmodel' % (a, b))

    def installed_languages(self):
        class Lang:
            def __init__(self, code):
                self.code = code

            def get_translation(self, to):
                frm = self.code
                return types.SimpleNamespace(translate=lambda s, frm=frm, to=to.code: '[%s>%s]%s' % (frm, to, s))
        return [Lang(c) for c in ('en', 'it', 'de', 'ru', 'ja', 'pt')]


def test_translator_pair_search_pivot_and_chain(golden_dir):
    gold = json.load(open(os.path.join(golden_dir, 't2t.json')))
    for rec in gold['translator']:
        be = ScriptedBackend(rec['pairs'])
        flt = (lambda text, from_code, to_code, tr: '<%s-%s>' % (from_code, to_code) + tr(text)) if rec['filter'] else None
        if 'raises' in rec:
            with pytest.raises(StopIteration):
                Translator(rec['from'], rec['to'], filter=flt, backend=be)
        else:
            t = Translator(rec['from'], rec['to'], filter=flt, backend=be)
            assert len(t.translators) == rec['nstages'] and t.translate('hello') == rec['out'], rec['name']
        assert be.log == rec['log'], rec['name']


def test_numbers_to_words_regex_suffixes_and_cache(golden_dir):
    gold = json.load(open(os.path.join(golden_dir, 't2t.json')))
    for rec in gold['numbers']:
        calls = []
        tr = None if rec['lang'] == 'en' else (lambda s: (calls.append(s), '{%s:%s}' % (rec['lang'], s))[1])
        n2w = NumbersToWords(rec['lang'], number_to_words=lambda s: 'N(%s)' % s, translator=tr)
        assert [n2w(t) for t in rec['texts']] == rec['out']
        assert calls == rec['translated']                    # one translation per distinct number string (the cache)


def test_english_number_words():
    w = english_number_to_words
    assert w('3') == 'three' and w('0') == 'zero' and w('115') == 'one hundred and fifteen'
    assert w('2999') == 'two thousand, nine hundred and ninety-nine' and w('30000') == 'thirty thousand'
    assert w('3,090.6') == 'three thousand and ninety point six' and w('29.0') == 'twenty-nine point zero'
    assert w('21,188,128') == 'twenty-one million, one hundred and eighty-eight thousand, one hundred and twenty-eight'
    assert w('1000001') == 'one million and one' and w('1100') == 'one thousand, one hundred'
    assert NumbersToWords(number_to_words=w)('I have 50% cats and 2 dogs.') == 'I have fifty percent cats and two dogs.'
    with pytest.raises(ValueError):
        w('12a')


def test_globals_translator_is_cached(monkeypatch):
    from infernos_amd import t2t, torcher
    made = []

    class FakeTranslator:
        def __init__(self, a, b, **kw):
            made.append((a, b))
    monkeypatch.setattr(t2t, 'Translator', FakeTranslator)
    torcher.InfernGlobals.get_translator.cache_clear()
    a = torcher.InfernGlobals.get_translator('en', 'de')
    b = torcher.InfernGlobals.get_translator('en', 'de')
    c = torcher.InfernGlobals.get_translator('en', 'it')
    assert a is b and a is not c and made == [('en', 'de'), ('en', 'it')]
    torcher.InfernGlobals.get_translator.cache_clear()
