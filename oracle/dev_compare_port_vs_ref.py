"""oracle/dev_compare_port_vs_ref.py -- TEST INFRASTRUCTURE ONLY (development aid, runs only where /root/reference exists):
times and compares the C port against the real reference decoder on ad-hoc cases; the committed form of this check is
tests/test_oracle_port.py and tests/test_oracle_fuzz.py."""
import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from oracle import ref, port
REF='/root/reference'
A = ref.Alphabet(REF+'/data/alphabet.txt'); labels, space = port.parse_alphabet_file(REF+'/data/alphabet.txt')
S = ref.Scorer(REF+'/data/smoke_test/pruned_lm.scorer', A); P = port.Scorer(REF+'/data/smoke_test/pruned_lm.scorer')
AU = ref.Alphabet(None); ulabels, uspace = port.utf8_alphabet()
SU = ref.Scorer(REF+'/data/smoke_test/pruned_lm.bytes.scorer', AU); PU = port.Scorer(REF+'/data/smoke_test/pruned_lm.bytes.scorer')
print('bytes scorer', PU.utf8, PU.order, PU.model_type, PU.alpha, PU.beta, SU.alpha, SU.beta)
vocab = open(REF+'/data/smoke_test/vocab.pruned.txt').read().split()
def canon(res):
    return sorted([(float(c), tuple(t.tolist()), tuple(ts.tolist())) for c,t,ts in res])
def cmp(name, dr, dp, p, chunks=None, nres=None):
    if chunks:
        for i in range(0, len(p), chunks): dr.next(p[i:i+chunks]); dp.next(p[i:i+chunks])
    else:
        dr.next(p); dp.next(p)
    n = nres or dp.beam
    rr, rp = dr.decode(n), dp.decode(n)
    ok_exact = len(rr)==len(rp) and all(a[0]==b[0] and np.array_equal(a[1],b[1]) and np.array_equal(a[2],b[2]) for a,b in zip(rr,rp))
    ok_canon = canon(rr)==canon(rp)
    top = rr[0][0]==rp[0][0] and np.array_equal(rr[0][1], rp[0][1])
    print('%-40s n=%d exact=%s canon=%s top1=%s' % (name, len(rp), ok_exact, ok_canon, top))
    return ok_canon
rng = np.random.RandomState(7)
def peaky_labels(lab, T, Cn, blank, rng, noise=0.02, hold=2):
    plan = [blank]*int(rng.randint(5,25))
    for l in lab: plan += [l]*hold + [blank]*int(rng.randint(1,3))
    plan = (plan + [blank]*T)[:T]
    p = noise*rng.rand(T,Cn)
    for t,c in enumerate(plan): p[t,c] += 0.6+0.35*rng.rand()
    return p/p.sum(1,keepdims=True)
allok = True
for it in range(6):
    sent = ' '.join(rng.choice(vocab, size=rng.randint(3,9)))
    lab = [0 if ch==' ' else ord(ch)-ord('a')+1 if ch!="'" else 27 for ch in sent]
    T = min(400, 30 + 5*len(lab))
    noise = [0.02, 0.1, 0.3, 1.0, 0.02, 0.2][it]
    p = peaky_labels(lab, T, 29, 28, rng, noise)
    for beam, use in [(100, False), (100, True), (500, True)]:
        allok &= cmp('word it%d noise%.2f beam%d lm%d'%(it,noise,beam,use), ref.Decoder(A, beam, S if use else None), port.Decoder(labels, space, beam, P if use else None), p)
    allok &= cmp('word it%d stream16 beam64 lm' % it, ref.Decoder(A, 64, S), port.Decoder(labels, space, 64, P), p, chunks=16)
# uniform-ish softmax of random logits
p = rng.randn(120,29)*0.5; p = np.exp(p); p/=p.sum(1,keepdims=True)
allok &= cmp('random emissions beam200 nolm', ref.Decoder(A,200,None), port.Decoder(labels,space,200,None), p)
allok &= cmp('random emissions beam200 lm', ref.Decoder(A,200,S), port.Decoder(labels,space,200,P), p)
# hot words, cutoff_prob
sent='she had your dark suit'; lab=[0 if ch==' ' else ord(ch)-ord('a')+1 for ch in sent]
p = peaky_labels(lab, 150, 29, 28, rng, 0.3)
hw={'dark':5.0,'suit':-3.0}
allok &= cmp('hot words beam100', ref.Decoder(A,100,S,hot_words=hw), port.Decoder(labels,space,100,P,hot_words=hw), p)
allok &= cmp('cutoff_prob .95 top_n 10 beam100 lm', ref.Decoder(A,100,S,cutoff_prob=0.95,cutoff_top_n=10), port.Decoder(labels,space,100,P,cutoff_prob=0.95,cutoff_top_n=10), p)
allok &= cmp('cutoff_top_n 10 beam100 nolm', ref.Decoder(A,100,None,cutoff_prob=1.0,cutoff_top_n=10), port.Decoder(labels,space,100,None,cutoff_prob=1.0,cutoff_top_n=10), p)
# utf8 / bytes mode
for it in range(3):
    sent = ' '.join(rng.choice(vocab, size=rng.randint(2,5)))
    lab = [b-1 for b in sent.encode()]
    p = peaky_labels(lab, 30+5*len(lab), 256, 255, rng, [0.002,0.01,0.05][it])
    for beam,use in [(50,False),(100,True),(1024,True)]:
        allok &= cmp('bytes it%d beam%d lm%d'%(it,beam,use), ref.Decoder(AU,beam,SU if use else None), port.Decoder(ulabels,uspace,beam,PU if use else None), p, nres=min(beam,200))
print('ALL OK' if allok else 'SOME FAILED')
