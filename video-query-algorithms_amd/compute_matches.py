"""Drop-in for the reference's ``compute_matches`` (src/models/compute_matches.py:8-107).

The orchestration of one broker pass: every pending query update becomes a ticket, gets its target, is scanned,
re-weighted, scored and turned into the next review set.  The sequence of calls on the ticket is the reference's (so
the REST side effects of its Ticket methods happen in the same order); the arithmetic behind ``compute_similarities``,
``optimize_weights``, ``compute_scores`` and ``select_clips_to_review`` runs on the GPU (``TicketScoring``,
``Hyperparameter``, ``TargetClip``).  ``ticket_factory(update_object, url)`` supplies the Ticket class -- the reference's
own, patched by ``install``, or the offline ``Ticket``.

In production this module is NOT needed: after ``install()`` the reference's own ``compute_matches.py`` runs
unmodified on the patched classes.  It exists as the offline harness -- a broker pass with no REST
server (tests, ``tools/e2e_cfg5.py``): with no ``ticket_factory`` every update object carries its own search set
(``records`` and/or ``feature_db``) and the side effects land in ``Ticket.ledger``.
"""
from __future__ import annotations

import os

from .target_clip import TargetClip

_IN_PROGRESS, _PROCESSED, _ERROR, _FINALIZED = 3, 4, 5, 7      # process states, compute_matches.py:42,50,104,107


def _choose_weights(kind, update, tk, hp):
    """compute_matches.py:60-67: defaults for a new query (or one without matches yet), else re-fit on the labels."""
    if kind == "new" or not update["matches"]:
        hp.weights = hp.default_weights
        hp.threshold = hp.default_threshold
    elif kind in ("revise", "finalize"):
        hp.optimize_weights(tk)
    else:
        raise Exception('update type is invalid')


def _review_budget(kind, tk, hp):
    """compute_matches.py:78-88: (how many matches go out, how far below the threshold to look).  A finalize pass
    reports everything and reaches down to the lowest-scoring match the user confirmed."""
    if kind != "finalize":
        return tk.number_of_matches_to_review, hp.near_miss_default
    lowest, _clip = tk.lowest_scoring_user_match()
    reach = max(hp.threshold - lowest, 0) / max(1 - hp.threshold, float(os.environ["COMPUTE_EPS"]))
    return float("inf"), reach


def _one_update(kind, update, url, hp, ticket_factory, target_factory):
    tk = ticket_factory(update, url)
    tk.change_process_state(_IN_PROGRESS)
    fatal, warning = tk.catch_errors(kind)                                  # :47-52
    if fatal:
        tk.change_process_state(_ERROR, message=fatal)
        return
    if warning:
        tk.add_note(warning)

    tk.target = target_factory(tk, hp)                                      # :55-58
    tk.target.get_target_features()
    tk.compute_similarities(hp)
    _choose_weights(kind, update, tk, hp)

    round_no = 1 if kind == 'new' else tk.latest_query_result["round"] + 1  # :70-74
    result_id = tk.create_query_result(round_no, hp)

    tk.compute_scores(hp.weights)                                           # :77
    budget, reach = _review_budget(kind, tk, hp)
    tk.select_clips_to_review(hp.threshold, budget, reach)
    if not tk.matches:                                                      # :92-94
        catch_no_matches_error(tk)
        return
    tk.add_matches_to_database(result_id)                                   # :97
    if kind == "finalize":                                                  # :102-107
        tk.create_final_report(hp, result_id)
        tk.change_process_state(_FINALIZED)
    else:
        tk.change_process_state(_PROCESSED)


def compute_matches(query_updates, hyperparameters, ticket_factory=None, target_factory=TargetClip):
    pending = query_updates.get_status()                                    # :34
    if ticket_factory is None:
        from .ticket import Ticket as _Ticket

        def ticket_factory(update_object, url):
            # offline: the update object itself carries the search set (API feature records and/or a resident
            # FeatureDB); the REST side effects of the round are kept in the ticket's ledger
            return _Ticket(update_object, records=update_object.get("records"), feature_db=update_object.get("feature_db"))
    for kind, update in pending.items():                                    # :37
        if update is not None:
            _one_update(kind, update, query_updates.url, hyperparameters, ticket_factory, target_factory)


def catch_no_matches_error(ticket):
    """compute_matches.py:110-114."""
    which = ticket.latest_query_result["round"] if ticket.latest_query_result else 1
    ticket.change_process_state(_ERROR, message="*** Error: No matches were found for round {} of query {}! ***".format(
        which, ticket.query_id))
