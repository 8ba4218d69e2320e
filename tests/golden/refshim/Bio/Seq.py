class Seq: pass
class PairwiseAligner: pass
