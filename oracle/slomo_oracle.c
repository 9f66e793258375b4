/* slomo oracle: filled in below */
