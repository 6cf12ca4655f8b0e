from .box_voting_head import BoxVotingHead
from .centroids_voting_head import CentroidVotingHead
from .voting_head_template import VotingHeadTemplate

__all__ = {
    'VotingHeadTemplate': VotingHeadTemplate,
    'CentroidVotingHead': CentroidVotingHead,
    'BoxVotingHead': BoxVotingHead,
}
