"""Drop-in for the reference's ``TargetClip`` (src/models/target_clip.py): the query vectors of a round.

* no bootstrapping (target_clip.py:37-39, :51-53): ``t = r / (r . r)`` of the reference clip (SURVEY.md 8(a) row B1:
  ``_get_clip_features`` :263-286, ``scaled_ref_clip_features`` :137-143, ``_scale_feature`` :311-313) -- a few
  thousand numbers, host numpy as in the reference (``FeatureDB.set_query_from_row`` is the on-device variant);
* dynamic target adjustment (:41-73): 'simple', 'partial_update' and 'bagging' keep the reference's
  control flow, its calls into ``random`` (same draws for the same seed) and its numpy averaging on the host; the matrix
  formulas (:192-197, :245-260) of all (stream, split[, bag]) problems of the round run in one GPU launch
  (``bootstrap.bootstrap_targets`` -> csrc/vq_boot.hip) instead of 1024 x 1024 host inverses.
"""
from __future__ import annotations

import random

import numpy as np

from .bootstrap import bootstrap_targets


class TargetClip:
    def __init__(self, ticket, hyperparameters):
        # target_clip.py:9-24
        self.client = getattr(ticket, "client", None)
        self.schema = getattr(ticket, "schema", None)
        self._ticket = ticket
        self.bootstrap_target = ticket.dynamic_target_adjustment
        self.latest_query_result = ticket.latest_query_result
        self.hyperparameters = hyperparameters
        self.ref_clip_features, self.splits = self._get_clip_features(ticket.ref_clip_id)
        self.previous_target_features = None
        self.target_features = {}
        if ticket.latest_query_result:
            if ticket.latest_query_result["bootstrapped_target"]:
                self.previous_target_features = ticket.latest_query_result["bootstrapped_target"]

    def get_target_features(self):
        """target_clip.py:26-73."""
        if not self.bootstrap_target or self.latest_query_result is None:                     # case 1
            self.target_features = self.scaled_ref_clip_features()
            return
        features_4_matches, splits_4_matches = self.features_for_matches(user_match_value=True)
        features_invalid_matches, __ = self.features_for_matches(user_match_value=False)
        if not features_4_matches:                                                            # case 2
            self.target_features = self.scaled_ref_clip_features()
            return
        hp = self.hyperparameters
        if hp.bootstrap_type == "simple":                                                     # case 3
            self.target_features = self.dynamic_target_adjustment(features_4_matches, features_invalid_matches,
                                                                  splits_4_matches, hp.f_bootstrap, replacement=False)
        elif hp.bootstrap_type == "partial_update":                                           # case 4
            self.target_features = self.dynamic_target_adjustment(features_4_matches, features_invalid_matches,
                                                                  splits_4_matches, hp.f_bootstrap, replacement=False)
            self.avg_new_old_targets(splits_4_matches)
        elif hp.bootstrap_type == "bagging":                                                  # case 5
            self.target_by_bagging(features_4_matches, features_invalid_matches, splits_4_matches)
        else:
            raise Exception("Error: bootstrap_type should be one of 'simple', 'partial_update', or 'bagging'")

    def avg_new_old_targets(self, splits):
        """target_clip.py:75-82.  (The reference leaves ndarrays here, which its own json.dumps of the target at
        ticket.py:296 cannot serialise; lists are stored instead -- same values.)"""
        if not self.previous_target_features:
            return
        hp = self.hyperparameters
        for stream in hp.streams:
            for split in splits:
                prev = self._previous(stream, split)
                self.target_features[stream][split] = (np.multiply(hp.f_memory, self.target_features[stream][split])
                                                       + np.multiply((1 - hp.f_memory), prev)).tolist()

    def _previous(self, stream, split):
        d = self.previous_target_features[stream]
        return d[split] if split in d else d[str(split)]       # a target that went through JSON has string keys

    def dynamic_target_adjustment(self, list_of_feature_dictionaries, list_invalid_feature_dicts, splits, b_fraction,
                                  replacement=False):
        """target_clip.py:84-104: one new target {stream: {split: list}}."""
        return self._solve([self._draw(list_of_feature_dictionaries, list_invalid_feature_dicts, b_fraction, replacement)],
                           splits)[0]

    def target_by_bagging(self, features_4_matches, features_invalid_matches, splits):
        """target_clip.py:145-159: nbags resamples with replacement, then the mean of the bag targets."""
        hp = self.hyperparameters
        draws = [self._draw(features_4_matches, features_invalid_matches, 1, True) for _ in range(hp.nbags)]
        bags = self._solve(draws, splits)
        self.target_features = {}
        for stream in hp.streams:
            self.target_features[stream] = {}
            for split in splits:
                self.target_features[stream][split] = np.average([bags[b][stream][split] for b in range(hp.nbags)],
                                                                 axis=0).tolist()

    def _draw(self, valid, invalid, b_fraction, replacement):
        """The random selections of one target, in the reference's order (target_clip.py:181-183 / :227-230)."""
        if invalid:
            valid = self._random_fraction(valid, b_fraction, replacement)
            invalid = self._random_fraction(invalid, b_fraction, replacement)
            return valid, invalid
        if b_fraction != 1 or replacement is True:
            valid = self._random_fraction(valid, b_fraction, replacement)
        return valid, []

    def _solve(self, draws, splits):
        """All (draw, stream, split) problems in one device launch -> [{stream: {split: list}}] per draw."""
        hp = self.hyperparameters
        problems, index = [], []
        for di, (valid, invalid) in enumerate(draws):
            xf = self._stack(valid, splits)
            yf = self._stack(invalid, splits) if invalid else None
            for stream in hp.streams:
                for split in splits:
                    problems.append((np.asarray(xf[stream][split], dtype=np.float64),
                                     np.asarray(yf[stream][split], dtype=np.float64) if yf else None))
                    index.append((di, stream, split))
        out = bootstrap_targets(problems, hp.mu, device=getattr(getattr(self, "_ticket", None), "device", 0) or 0)
        targets = [{stream: {} for stream in hp.streams} for _ in draws]
        for (di, stream, split), w in zip(index, out):
            targets[di][stream][split] = w.tolist()
        return targets

    def _stack(self, dicts, splits):
        """target_clip.py:186-190 / :233-242: per (stream, split) the feature vectors in list order."""
        out = {stream: {split: [] for split in splits} for stream in self.hyperparameters.streams}
        for feature_dictionary in dicts:
            for stream_type, split_features in feature_dictionary.items():
                for split, feature in split_features.items():
                    out[stream_type][split].append(feature)
        return out

    def features_for_matches(self, user_match_value=True):
        """target_clip.py:106-135."""
        page = 1
        matches = []
        while page is not None:
            results = self._request(["matches", "list"], {"query_result": self.latest_query_result["id"], "page": page})
            matches.extend(results["results"])
            page = results["pagination"]["nextPage"]
        matches_features = []
        splits_matches = set()
        for match in matches:
            if match["user_match"] is user_match_value:
                match_features, match_splits = self._get_clip_features(match["video_clip"])
                matches_features.append(match_features)
                splits_matches.update(match_splits)
        return matches_features, splits_matches

    def scaled_ref_clip_features(self):
        """target_clip.py:137-143."""
        ref_features = {}
        for stream, split_features in self.ref_clip_features.items():
            ref_features[stream] = {}
            for split, feature in split_features.items():
                ref_features[stream][split] = self._scale_feature(feature).tolist()
        return ref_features

    def _get_clip_features(self, clip_id):
        """target_clip.py:263-286."""
        results = {stream_type: {} for stream_type in self.hyperparameters.streams}
        splits = set()
        for feature_object in self._request(["video-clips", "features"], {"id": clip_id}):
            stream_type = feature_object["dnn_stream_id"]
            if stream_type in self.hyperparameters.streams and feature_object["name"] == self.hyperparameters.feature_name:
                fsplit = feature_object["dnn_stream_split"]
                splits.add(fsplit)
                results[stream_type][fsplit] = feature_object["feature_vector"]
        return results, splits

    def _request(self, action, params):
        return self._ticket._request(action, params)

    @staticmethod
    def _random_fraction(flist, fraction, replacement):
        """target_clip.py:297-309 (same calls into ``random``, same ``list(set(...))`` order)."""
        nmatches = len(flist)
        tmatches = round(nmatches * fraction)
        tmatches = max(tmatches, 1)
        if replacement is False:
            tsamples = random.sample(range(nmatches), tmatches)
        else:
            tsamples = random.choices(range(nmatches), k=tmatches)
        tsamples = list(set(tsamples))
        return [flist[m] for m in tsamples]

    @staticmethod
    def _scale_feature(f):
        """target_clip.py:311-313."""
        return f / np.dot(f, f)
