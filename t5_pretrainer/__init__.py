"""Import-path alias so the reference's shell pipeline (full_scripts/full_evaluate_t5seq_aq_encoder.sh:
176-205: ``python -m t5_pretrainer.evaluate --task=t5seq_aq_retrieve_docids[_2]``,
``python -m t5_pretrainer.aq_preprocess.build_list_smtid_to_nextids``) and code importing
``t5_pretrainer.tasks.generation`` / ``t5_pretrainer.modeling.t5_generative_retriever`` run
unmodified on the MI355X implementation in ``ripor_amd``. Only the generative-retrieval path exists."""
