class SeqRecord: pass
class PairwiseAligner: pass
