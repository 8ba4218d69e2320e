"""Minimal stand-in for Biopython: only SeqIO.parse(fasta) is functional (golden generation only)."""
