"""Agent registry (reference: src/modules/agents/__init__.py:1-13). Only the entity-attention
recurrent agents -- the ones on the hot path of algs/refil.yaml and algs/qmix_atten.yaml -- exist here;
flat-observation agents and the feed-forward twins are out of scope this round."""
REGISTRY = {}

from .entity_rnn_agent import EntityAttentionRNNAgent, ImagineEntityAttentionRNNAgent

REGISTRY["entity_attend_rnn"] = EntityAttentionRNNAgent
REGISTRY["imagine_entity_attend_rnn"] = ImagineEntityAttentionRNNAgent
