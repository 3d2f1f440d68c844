"""Text-to-text stage of the LiveTranslator / attendant paths: Core/T2T/Translator.py:19-57 and
Core/T2T/NumbersToWords.py:7-35.  Host text code; the engines behind them are third-party packages the reference
installs (`argostranslate`, `inflect`) and this image does not carry, so both classes take the engine as an argument
and fall back to the package when it is importable:

  * Translator(from_code, to_code, filter=None, backend=None): direct language pair if the package index has one,
    otherwise a pivot through one of `supported_langs` (tried from the END of the list, as the reference's `pop()` does);
    `translate` chains the stages.  `backend` needs `load_pair(from_code, to_code)` (raises StopIteration when the index
    has no such pair) and `installed_languages()` -> objects with `.code` and `.get_translation(to_lang).translate`.
  * NumbersToWords(lang='en', number_to_words=None, translator=None): every number in a text (the reference's regex)
    replaced by its words, '%' -> ' percent', trailing '.', ',', '!' kept, translated and cached for other languages.
    Without `inflect`, `english_number_to_words` spells the number (inflect's conventions: hyphenated tens, 'and' after
    hundreds, commas between groups, 'point' + digits) -- its output against inflect's is unpinned.
"""
import re
from functools import partial
from typing import Optional, Tuple


class ArgosBackend:
    """the calls Core/T2T/Translator.py:8-17,44 makes into argostranslate"""

    def __init__(self):
        import argostranslate.package  # noqa: F401  (ImportError here: pass backend= instead)
        import argostranslate.translate  # noqa: F401

    def load_pair(self, from_code, to_code):
        import argostranslate.package
        argostranslate.package.update_package_index()
        available = argostranslate.package.get_available_packages()
        pkg = next(filter(lambda x: x.from_code == from_code and x.to_code == to_code, available))
        argostranslate.package.install_from_path(pkg.download())

    def installed_languages(self):
        from argostranslate.translate import get_installed_languages
        return get_installed_languages()


class Translator():
    supported_langs = ["en", "it", "de", "ru", "ja"]
    translators: Tuple[callable]

    def __init__(self, from_code: str, to_code: str, filter: Optional[callable] = None, backend=None):
        be = backend if backend is not None else ArgosBackend()
        langs = None
        stages, cur = [], from_code
        for code in self._plan(be, from_code, to_code):
            if langs is None:
                langs = {lang.code: lang for lang in be.installed_languages()}
            step = langs[cur].get_translation(langs[code]).translate
            stages.append(step if filter is None else partial(filter, from_code=cur, to_code=code, tr=step))
            cur = code
        self.translators = tuple(stages)

    def _plan(self, be, src: str, dst: str):
        """The target codes of the stages, in order, after installing the language pairs they need: [dst] when the index
        has src -> dst; otherwise the first pivot, tried from the end of supported_langs, for which both src -> pivot and
        pivot -> dst install -- and then, as the reference does (Translator.py:38 queues the target, not the pivot), the
        plan is [dst, dst].  No pair and no pivot: the index's StopIteration propagates."""
        def install(a, b):
            try:
                be.load_pair(a, b)
            except StopIteration:
                return False
            return True
        if install(src, dst):
            return [dst]
        for via in reversed([c for c in self.supported_langs if c not in (src, dst)]):
            if install(src, via) and install(via, dst):
                return [dst, dst]
        raise StopIteration

    def translate(self, sourceText):
        text = sourceText
        for stage in self.translators:
            text = stage(text)
        return text


_ONES = ('zero one two three four five six seven eight nine ten eleven twelve thirteen fourteen fifteen sixteen seventeen '
         'eighteen nineteen').split()
_TENS = ('', '', 'twenty', 'thirty', 'forty', 'fifty', 'sixty', 'seventy', 'eighty', 'ninety')
_GROUPS = ('', ' thousand', ' million', ' billion', ' trillion', ' quadrillion', ' quintillion', ' sextillion', ' septillion',
           ' octillion', ' nonillion', ' decillion')


def _below_1000(n, use_and):
    out = []
    h, r = divmod(n, 100)
    if h:
        out.append(_ONES[h] + ' hundred')
    if r:
        words = _ONES[r] if r < 20 else (_TENS[r // 10] + ('-' + _ONES[r % 10] if r % 10 else ''))
        out.append(('and ' if (h or use_and) else '') + words)
    return ' '.join(out)


def english_number_to_words(num) -> str:
    """'3,090.6' -> 'three thousand and ninety point six' (separating commas dropped; digits after the point spelt singly)"""
    s = str(num).replace(',', '').strip()
    neg = s.startswith('-')
    s = s.lstrip('+-')
    whole, _, frac = s.partition('.')
    whole = whole or '0'
    if not whole.isdigit() or (frac and not frac.isdigit()):
        raise ValueError('not a number: %r' % (num,))
    n = int(whole)
    if n == 0:
        words = 'zero'
    else:
        groups = []
        while n:
            n, g = divmod(n, 1000)
            groups.append(g)
        if len(groups) > len(_GROUPS):
            raise ValueError('number too large to spell: %r' % (num,))
        parts = []
        for i in range(len(groups) - 1, -1, -1):
            if groups[i]:
                # the last group takes 'and' when a higher group precedes it and it is below one hundred
                parts.append(_below_1000(groups[i], use_and=(i == 0 and len(parts) > 0 and groups[i] < 100)) + _GROUPS[i])
        words = parts[0]
        for p in parts[1:]:
            words += (' ' if p.startswith('and ') else ', ') + p
    if frac:
        words += ' point ' + ' '.join(_ONES[int(c)] for c in frac)
    return ('minus ' if neg else '') + words


class NumbersToWords:
    tr: Optional[callable]
    cache: dict

    def __init__(self, lang='en', number_to_words=None, translator=None):
        if number_to_words is None:
            try:
                import inflect
                number_to_words = inflect.engine().number_to_words
            except ImportError:
                number_to_words = english_number_to_words
        self.number_to_words = number_to_words
        if lang == 'en':
            self.tr, self.cache = None, None
        else:
            if translator is None:
                from .torcher import InfernGlobals
                translator = InfernGlobals.get_translator('en', lang).translate
            self.tr, self.cache = translator, {}

    _NUMBER = re.compile(r'\b\d[\d.,]*%?(?=[\s.,!]|$)')           # the reference's pattern (NumbersToWords.py:18)

    def _spell(self, token: str) -> str:
        """one matched number: '%' reads ' percent', a trailing '.', ',' or '!' is punctuation and stays; other languages
        get the translated words, remembered per matched string"""
        if token.endswith('%'):
            body, tail = token[:-1], ' percent'
        elif token[-1] in '.,!':
            body, tail = token[:-1], token[-1]
        else:
            body, tail = token, ''
        words = self.number_to_words(body) + tail
        if self.tr is None:
            return words
        known = self.cache.get(token)
        if known is None:
            known = self.cache[token] = self.tr(words)
        return known

    def __call__(self, text):
        # every match replaces the FIRST occurrence of its string in the text as it stands by then (the reference's
        # str.replace(..., 1) semantics, which the fixture pins)
        for token in self._NUMBER.findall(text):
            text = text.replace(token, self._spell(token), 1)
        return text
