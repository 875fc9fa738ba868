from ripor_amd.aq_preprocess.build_list_smtid_to_nextids import main

if __name__ == "__main__":
    main()
