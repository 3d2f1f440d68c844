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
        to_code_p = [to_code, ]
        inter_codes = [x for x in self.supported_langs if x not in (from_code, to_code)]
        success = False
        while not success:
            try:
                be.load_pair(from_code, to_code)
            except StopIteration:
                pass
            else:
                success = True
                break
            while len(inter_codes) > 0:
                inter_code = inter_codes.pop()
                try:
                    be.load_pair(from_code, inter_code)
                    be.load_pair(inter_code, to_code)
                except StopIteration:
                    if len(inter_codes) == 0:
                        raise
                    continue
                # NB: the reference inserts to_code (not inter_code) in front, so a pivoted chain asks the installed
                # languages for from -> to twice (Translator.py:38); kept as is
                to_code_p.insert(0, to_code)
                success = True
                break
        ilangs = dict((x.code, x) for x in be.installed_languages())
        from_lang = ilangs[from_code]
        translators = []
        for tc in to_code_p:
            to_lang = ilangs[tc]
            tr = from_lang.get_translation(to_lang).translate
            if filter is not None:
                tr = partial(filter, from_code=from_code, to_code=tc, tr=tr)
            translators.append(tr)
            from_lang, from_code = to_lang, tc
        self.translators = tuple(translators)

    def translate(self, sourceText):
        for translator in self.translators:
            sourceText = translatedText = translator(sourceText)
        return translatedText


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

    def __call__(self, text):
        numbers = re.findall(r'\b\d[\d.,]*%?(?=[\s.,!]|$)', text)
        for number in numbers:
            if number.endswith('%'):
                tr_number = number[:-1]
                suffix = ' percent'
            elif number[-1] in ('.', ',', '!'):
                tr_number = number[:-1]
                suffix = number[-1]
            else:
                suffix = ''
                tr_number = number
            word = self.number_to_words(tr_number) + suffix
            if self.tr is not None:
                if (word_tr := self.cache.get(number, None)) is None:
                    self.cache[number] = word_tr = self.tr(word)
                word = word_tr
            text = text.replace(number, word, 1)
        return text
