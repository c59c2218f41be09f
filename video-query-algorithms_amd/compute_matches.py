"""Drop-in for the reference's ``compute_matches`` (src/models/compute_matches.py:8-107).

Same orchestration, same call order on the ticket (so the REST side effects of the reference's
Ticket methods happen in the same sequence); the arithmetic calls land on the GPU through
``TicketScoring`` / ``Hyperparameter``.  ``ticket_factory(update_object, url)`` lets the caller
supply the Ticket class to use (the reference's, patched by ``install``; or the offline ``Ticket``).
"""
from __future__ import annotations

import os

from .target_clip import TargetClip


def compute_matches(query_updates, hyperparameters, ticket_factory=None, target_factory=TargetClip):
    updates_needed = query_updates.get_status()                                    # compute_matches.py:34
    if ticket_factory is None:
        from .ticket import Ticket as _Ticket

        def ticket_factory(update_object, url):
            return _Ticket(update_object)
    for update_type, update_object in updates_needed.items():                      # :37
        if update_object is None:
            continue
        ticket = ticket_factory(update_object, query_updates.url)
        ticket.change_process_state(3)
        fatal_error_message, error_message = ticket.catch_errors(update_type)     # :47-52
        if fatal_error_message:
            ticket.change_process_state(5, message=fatal_error_message)
            continue
        if error_message:
            ticket.add_note(error_message)

        ticket.target = target_factory(ticket, hyperparameters)                    # :55-58
        ticket.target.get_target_features()
        ticket.compute_similarities(hyperparameters)

        if (update_type == "new") or not update_object["matches"]:                 # :61-67
            hyperparameters.weights = hyperparameters.default_weights
            hyperparameters.threshold = hyperparameters.default_threshold
        elif update_type == "revise" or update_type == "finalize":
            hyperparameters.optimize_weights(ticket)
        else:
            raise Exception('update type is invalid')

        new_round = 1 if update_type == 'new' else ticket.latest_query_result["round"] + 1   # :70-74
        new_result_id = ticket.create_query_result(new_round, hyperparameters)

        ticket.compute_scores(hyperparameters.weights)                             # :77-89
        if update_type == "finalize":
            max_number_matches = float("inf")
            low_score, __ = ticket.lowest_scoring_user_match()
            near_miss = max(hyperparameters.threshold - low_score, 0) / \
                max(1 - hyperparameters.threshold, float(os.environ["COMPUTE_EPS"]))
        else:
            max_number_matches = ticket.number_of_matches_to_review
            near_miss = hyperparameters.near_miss_default
        ticket.select_clips_to_review(hyperparameters.threshold, max_number_matches, near_miss)

        if not ticket.matches:                                                     # :92-94
            catch_no_matches_error(ticket)
            continue
        ticket.add_matches_to_database(new_result_id)                              # :97
        if update_type == "finalize":                                              # :102-107
            ticket.create_final_report(hyperparameters, new_result_id)
            ticket.change_process_state(7)
            continue
        else:
            ticket.change_process_state(4)


def catch_no_matches_error(ticket):
    """compute_matches.py:110-114."""
    mround = ticket.latest_query_result["round"] if ticket.latest_query_result else 1
    error_message = "*** Error: No matches were found for round {} of query {}! ***".format(mround, ticket.query_id)
    ticket.change_process_state(5, message=error_message)
