/* placeholder, filled in with the sampler restatement */
