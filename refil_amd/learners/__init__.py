"""Learner registry (reference: src/learners/__init__.py:1-4)."""
from .q_learner import QLearner

REGISTRY = {}
REGISTRY["q_learner"] = QLearner
