"""Golden vectors of the tweet normaliser: input strings and the outputs of the reference's
`Tweet_Preprocessing.normalizeTweet` (preprocessing/text_processing.py:180-248), imported from /root/reference in the
build container (it imports cleanly; `emoji` is absent there, so emojis stay as they are).  Fixture = data only.

    python tests/golden/make_text_golden.py      # writes tests/golden/text_golden.json
"""
import importlib.util
import json
import os
import random

HERE = os.path.dirname(os.path.abspath(__file__))
FRAGS = ["@john", "@a_b:", "https://t.co/abc", "www.x.com", "HTTP://X.Y", "I", "can't", "cannot", "ain't", "don't", "won't", "we're", "it's", "I'm",
         "they'll", "he'd", "I've", "5", "p.m.", "a.m.", "p", ".", "m", "a", "9", "p.m", "a.m", "!!!", "?!", ":-)", ":D", "<3", "#wow", "#1", "…", "’", "“quoted”",
         "RT", "soooo", "goooood", "&amp;", "&lt;3", "<b>html</b>", "foo@bar.com", "+1 (555) 123-4567", "$20.50", "100%", "U.S.A.", "♥", "😂😂", "😂", "\n", "\t",
         "  ", "hello", "world", "CAN'T", "Cannot", "n't", "ca", "ai", "'s", "'m", "—", "e.g.", "co-op", "12:30", "3.14", "1,000", "x'll", "y'd", "z've", "www", "http",
         "@", "#", "'", "\"", "(", ")", "..", "...", "....", "a.m.p.m.", "P.M.", "mañana", "日本語", "ок"]
HAND = ["", " ", "@john Check https://t.co/abc www.x.com I can't believe it's 5 p.m. already!!! :-) #wow <3 ain't cannot don't we're",
        "RT @a_b: soooo goooood…  “quoted” it’s 9 a.m. & more http://x.y", "email me at foo@bar.com or call +1 (555) 123-4567 :D :P ♥ 😂😂",
        "hello\n\nworld\t tabs   spaces", "<b>html</b> &amp; entities &lt;3", "U.S.A. is #1!!! $20.50 100% a.m p.m", "cannot cannot  cannot", "can't", "ca n't ai n't",
        "it's 5 p . m . now", "meet at 5 p.m", "WWW.EXAMPLE.COM Http://a.b @@user @ user", "I'm you're he's she'll we'd they've", "don't won't shouldn't"]


def main():
    spec = importlib.util.spec_from_file_location("ref_text_processing", "/root/reference/preprocessing/text_processing.py")
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    tp = ref.Tweet_Preprocessing()
    assert tp.demojizer is None, "fixtures are for the emoji-less environment"
    rng = random.Random(30)
    cases = list(HAND)
    for _ in range(400):
        k = rng.randint(1, 14)
        sep = rng.choice([" ", " ", " ", "", "  "])
        cases.append(sep.join(rng.choice(FRAGS) for _ in range(k)))
    out = [{"in": c, "out": tp.normalizeTweet(c)} for c in cases]
    with open(os.path.join(HERE, "text_golden.json"), "w", encoding="utf-8") as f:
        json.dump(out, f, ensure_ascii=False, indent=0)
    print("wrote", len(out), "cases")


if __name__ == "__main__":
    main()
