"""l3embedding_amd -- MI355X-native L3-Net AVC training path (drop-in for the
model/train entry points of marl/l3embedding)."""
__version__ = '0.1.0'
