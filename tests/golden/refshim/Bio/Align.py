class Align: pass
class PairwiseAligner: pass
